"""Diagnostic: eager UNet calls of the frozen LDM-4 engine as a DDIM step issues them (for rocprofv3 --kernel-trace --stats / --pmc)."""
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
eng.ctx_r = eng.context_branches(c)          # the per-step launch set of the sampling loops (context vectors and
eng.emb_r = eng.emb_rows(t)                  # time-embedding rows precomputed; the batch is a guidance pair [x, x])
x = torch.cat([x[:B], x[:B]]).contiguous()
eng.cfg_pair = True
torch.cuda.synchronize()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
for _ in range(2):
    eng(x, t, c)
torch.cuda.synchronize()
print("MARK")
for _ in range(int(os.environ.get('N_CALLS', '10'))):
    eng(x, t, c)
torch.cuda.synchronize()
