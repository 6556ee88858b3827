"""Diagnostic: in-kernel cycle stamps of the 4-wave GEMM k_gemm_nt (libedadm_stamps.so, `make -C eda-dm_amd/csrc stamps`) on the
short-K projection shapes of LDM-4: wave 0 of every eighth workgroup -- setup, K-step waits, the rest of the main loop, epilogue."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EDADM_LIB_PATH"] = os.path.join(ROOT, "eda-dm_amd", "csrc", "libedadm_stamps.so")
os.environ["EDADM_GEMM_FORCE"] = "2"          # the 4-wave kernel for every shape
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops, lib
dev = torch.device("cuda", 0)
L = lib.load()


def case(M, N, K, mode, residual):
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, K), dtype=torch.int8, device=dev)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if residual else None
    oqp = torch.tensor([0.05, 128.0, 255.0, 0.0], device=dev)
    out = torch.empty(M, N, device=dev)
    if mode == 0:
        f = lambda: ops.qgemm_i8(a, w, M, N, K, sc, bs, out, residual=res)
    else:
        f = lambda: ops.qgemm_i8_q(a, w, M, N, K, sc, bs, mode, oqp, residual=res)
    buf = (ctypes.c_ulonglong * 8)()
    for _ in range(3):
        f(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); L.edadm_dbg_read(buf)
    n = max(buf[4], 1)
    mfma = (K // 32) * 6 * 32 * 2
    print("M=%6d N=%4d K=%4d mode=%d res=%d | %6.1f us | per wave cycles: setup=%5.0f (addresses %5.0f, issue %5.0f) waits=%6.0f compute=%6.0f epilogue=%6.0f total=%6.0f | "
          "MFMA-rate cycles (two workgroups per CU) %d" % (M, N, K, mode, residual, e0.elapsed_time(e1) * 1e3, buf[0] / n, buf[6] / n, buf[7] / n, buf[1] / n,
                                                          buf[2] / n, buf[3] / n, buf[5] / n, mfma))


for M, N, K in ((102400, 384, 384), (102400, 384, 1536), (25600, 576, 576), (6400, 960, 960)):
    for mode, residual in ((0, 0), (0, 1), (1, 0), (2, 0)):
        case(M, N, K, mode, residual)
