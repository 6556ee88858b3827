"""Diagnostic: a few launches of the dominant conv shape for rocprofv3 --pmc (not part of the product)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)
B, H, Cin, Cout = 100, 64, 192, 192
x = torch.randint(-128, 128, (B, H, H, Cin), dtype=torch.int8, device=dev)
w = torch.randint(-8, 9, (Cout, 9 * Cin), dtype=torch.int8, device=dev)
sc, bs = torch.rand(Cout, device=dev) * 1e-3, torch.randn(Cout, device=dev)
M = B * H * H
out = torch.empty(M, Cout, device=dev)
res = torch.randn(M, Cout, device=dev)
geom = ops.make_geom(B, H, H, Cin, H, H, 3, 3, 1, 1, False, -1)
for _ in range(5):
    ops.qgemm_i8(x, w, M, Cout, 9 * Cin, sc, bs, out, geom=geom, residual=res, rows_per_batch=H * H)
a = torch.randint(-128, 128, (8192, 8192), dtype=torch.int8, device=dev)
b = torch.randint(-8, 9, (8192, 8192), dtype=torch.int8, device=dev)
o2 = torch.empty(8192, 8192, device=dev)
s2, b2 = torch.ones(8192, device=dev), torch.zeros(8192, device=dev)
for _ in range(3):
    ops.qgemm_i8(a, b, 8192, 8192, 8192, s2, b2, o2)
torch.cuda.synchronize()
