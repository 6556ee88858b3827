"""Diagnostic: A/B of Engine switches on the eager UNet call of the headline config (same box, same process):
python tools/call_ab.py gn_partials_gemm fused_split ...  -> ms per call with each named attribute True / False, alternating."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch, bench
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev)
eng = qnn.freeze()
B = 50
x = torch.randn(2 * B, 3, 64, 64, device=dev); t = torch.full((2 * B,), 501, dtype=torch.long, device=dev)
c = torch.randn(2 * B, 1, 512, device=dev)
eng.ctx_r = eng.context_branches(c)
eng.emb_r = eng.emb_rows(t)
x = torch.cat([x[:B], x[:B]]).contiguous()
eng.cfg_pair = True


def ms(n=10):
    for _ in range(2):
        y = eng(x, t, c)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = eng(x, t, c)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y.clone()


for name in sys.argv[1:]:
    assert hasattr(eng, name), name
    ys = {}
    for rep in range(2):
        for v in (True, False):
            setattr(eng, name, v)
            m, y = ms()
            ys[v] = y
            print("%s=%s: %.3f ms per call" % (name, v, m), flush=True)
    setattr(eng, name, True)
    d = (ys[True] - ys[False]).abs()
    print("%s: outputs differ on %d of %d values, max %.3e (range %.3e)" % (name, int((d > 0).sum()), d.numel(), float(d.max()), float(ys[True].abs().max())))
