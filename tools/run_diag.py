import ctypes, torch, sys, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemm_diag.so"))
dev = torch.device("cuda", 0)
def run(M, N, K, geom=None, x=None, w=None, res=False, label=""):
    sc, bs = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    lib.edadm_qgemm_i8.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_int64]*3 + [ctypes.c_void_p]*4 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    for it in range(3):
        rc = lib.edadm_qgemm_i8(P(x), K, P(w), K, M, N, K, ctypes.cast(geom, ctypes.c_void_p) if geom is not None else None, P(sc), P(bs), None, 1, P(r), N, P(out), N, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        lib.edadm_dbg_read(buf)
    n = buf[4]
    print("%s waves=%d per-wave cycles: total_main=%.0f wait_vmcnt=%.0f barrier=%.0f issue=%.0f epilogue=%.0f" % (label, n, buf[5]/n, buf[0]/n, buf[1]/n, buf[2]/n, buf[3]/n))
B,H,C = 100,64,192
x = torch.randint(-128,128,(B,H,H,C),dtype=torch.int8,device=dev); w = torch.randint(-8,9,(192,9*C),dtype=torch.int8,device=dev)
geom = (ctypes.c_int32*16)(1,B,H,H,C,H,H,3,3,1,1,0,-1,0,0,0)
run(B*H*H, 192, 9*C, geom, x, w, label="conv192@64 K=1728")
run(B*H*H, 192, 9*C, geom, x, w, res=True, label="conv192@64 +res   ")
a = torch.randint(-128,128,(409600,1728),dtype=torch.int8,device=dev)
run(409600, 192, 1728, None, a, w, label="dense 409600x192x1728")
a2 = torch.randint(-128,128,(8192,8192),dtype=torch.int8,device=dev); w2 = torch.randint(-8,9,(8192,8192),dtype=torch.int8,device=dev)
run(8192, 8192, 8192, None, a2, w2, label="dense 8192^3")
