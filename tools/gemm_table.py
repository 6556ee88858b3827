"""Diagnostic (not part of the product): per-layer timing table of one eager UNet call."""
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
for _ in range(2):
    eng(x, t, c)
torch.cuda.synchronize()
eng.prof = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng(x, t, c); e1.record(); torch.cuda.synchronize()
print("eager unet ms", e0.elapsed_time(e1))
rows = {}
def kms(run, reps=5):
    run(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): run()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps
prof, eng.prof = eng.prof, None
for mode, name, M, N, K, f, run in prof:
    key = (mode, M, N, K)
    r = rows.setdefault(key, [0, 0.0, 0.0])
    r[0] += 1; r[1] += kms(run); r[2] += f
tot = sum(r[1] for r in rows.values())
print("gemm total ms", tot)
for key, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-4s M=%7d N=%5d K=%5d  n=%3d  ms=%7.3f  TF/s=%7.1f" % (key[0], key[1], key[2], key[3], r[0], r[1], r[2] / r[1] / 1e9))
