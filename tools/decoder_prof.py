"""Diagnostic: the VQ-f4 first-stage decoder alone (for rocprofv3 --stats + tools/prof_diff.py): CALLS decodes of a 16-image chunk."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
dec = bench.make_decoder(dev)
z = torch.randn(16, 3, 64, 64, device=dev)
for _ in range(int(os.environ.get("CALLS", "2"))):
    dec(z)
torch.cuda.synchronize()
