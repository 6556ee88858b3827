"""Diagnostic: a few launches of k_conv3_direct at the LDM-4 shapes for rocprofv3 --pmc (not part of the product)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
from edadm import ops
dev = torch.device("cuda", 0)
for B, H, Cin, N in ((100, 64, 576, 192), (100, 32, 384, 384), (100, 16, 576, 576), (100, 8, 960, 960)):
    a = torch.randint(-128, 128, (B, H, H, Cin), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 9, (N, 3, 3, Cin), dtype=torch.int8, device=dev)
    wdc = ops.conv3_pack_w(w, N, Cin)
    sc, bs = torch.rand(N, device=dev) * 1e-3, torch.randn(N, device=dev)
    out = torch.empty(B * H * H, N, device=dev)
    for _ in range(3):
        ops.qconv3_i8_direct(a, wdc, B, H, H, Cin, N, 0, sc, bs, out)
torch.cuda.synchronize()
