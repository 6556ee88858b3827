"""Diagnostic: eager FP32-reference UNet forwards of LDM-4 at 128 rows (a TDAC trajectory step with CFG) for rocprofv3 --kernel-trace --stats."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch, bench
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev, calib_rows=16)
qnn.set_quant_state(False, False)
B = 128
x = torch.randn(B, 3, 64, 64, device=dev); t = torch.full((B,), 501, dtype=torch.long, device=dev); c = torch.randn(B, 1, 512, device=dev)
with torch.no_grad():
    for _ in range(2): qnn(x, t, c)
    torch.cuda.synchronize(); t0 = time.time()
    n = int(os.environ.get("N_CALLS", "4"))
    for _ in range(n): y = qnn(x, t, c)
    torch.cuda.synchronize(); print("FP forward eager, %d rows: %.1f ms" % (B, (time.time() - t0) / n * 1e3))
