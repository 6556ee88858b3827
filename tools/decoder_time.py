"""Diagnostic: wall time of the VQ-f4 first-stage decoder (bench.time_decoder) with and without the fused GroupNorm expansion."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
dec = bench.make_decoder(dev)
if os.environ.get("CHUNK"):
    dec.chunk_pixels = int(os.environ["CHUNK"])
for fuse in (True, False, True):
    dec.fuse_gn_split = fuse
    r = bench.time_decoder(dec, dev, 50)
    print("fuse_gn_split", fuse, "ms/image %.4f" % r["ms_per_image"])
