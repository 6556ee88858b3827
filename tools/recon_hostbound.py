"""Diagnostic: is a reconstruction iteration of unit X bound by the GPU or by the host (Python / autograd dispatch)?
Wall time per iteration against the summed device time of its kernels (torch profiler), for units of every level."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev, calib_rows=16)
from qdiff.block_recon import block_reconstruction
import edadm.recon as er
g = torch.Generator().manual_seed(3)
N = 64
cali = (torch.randn(N, 3, 64, 64, generator=g).to(dev), torch.randint(1, 1000, (N,), generator=g).to(dev),
        torch.randn(N, 1, 512, generator=g).to(dev))
qnn.set_quant_state(True, True)
m = qnn.model
units = [("res 192@64", m.input_blocks[1][0]), ("tf 384@32", m.input_blocks[4][1].transformer_blocks[0]),
         ("res 576@16", m.input_blocks[7][0]), ("tf 576@16", m.input_blocks[7][1].transformer_blocks[0]),
         ("res 960@8", m.middle_block[0]), ("tf 960@8", m.middle_block[1].transformer_blocks[0])]
kw = dict(cali_data=cali, act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-1, p=2.0, weight=0.0001, b_range=(20, 2),
          warmup=0.2, batch_size=32, input_prob=0.5, add_loss=0.8, recon_w=True, recon_a=True, keep_gpu=True)
for name, unit in units:
    block_reconstruction(qnn, unit, iters=3, **kw)          # caches, code objects
    er.TIMING = {"iter_s": 0.0, "iters": 0}
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        block_reconstruction(qnn, unit, iters=11, **kw)
        torch.cuda.synchronize()
    t = er.TIMING
    er.TIMING = None
    dev_us = sum(getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0) for e in prof.key_averages())
    n_k = sum(e.count for e in prof.key_averages())
    print("%-12s wall %.2f ms/iter | device %.2f ms/iter (all 11 iters + caching: upper bound) | %d kernels/iter" % (
        name, 1e3 * t["iter_s"] / t["iters"], dev_us / 1e3 / 11, n_k // 11))
