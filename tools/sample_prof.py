"""Diagnostic: ONE eager (no HIP graph: every launch visible to rocprofv3) DDIM sample of a 50-image batch on the frozen LDM-4 engine --
20 steps x CFG 3.0 = 20 UNet calls of 100 rows + 20 edadm_ddim_step launches -- twice (the first untimed warm-up is part of the profile
too: tools/elementwise_hbm.py works on whole-process totals).  For tools/prof_elementwise.sh."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
from edadm.sampling import DDIMLoop
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev)
eng = qnn.freeze()
B = 50
loop = DDIMLoop(eng, (3, 64, 64), B, steps=20, eta=0.0, scale=3.0, context_shape=(1, 512), device=dev, use_graph=False)
g = torch.Generator(device=dev).manual_seed(1)
noise = torch.randn(B, 3, 64, 64, generator=g, device=dev)
cond = torch.randn(B, 1, 512, generator=g, device=dev)
uncond = torch.randn(1, 1, 512, generator=g, device=dev).expand(B, 1, 512).contiguous()
for _ in range(int(os.environ.get("SAMPLES", "2"))):
    out = loop.sample(noise, cond, uncond)
torch.cuda.synchronize()
print("SAMPLES done", tuple(out.shape))
