"""Diagnostic: the reconstruction inner loop alone (for rocprofv3 --stats): three full-size LDM-4 units, ITERS
iterations each at batch 32, activation caches built beforehand."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev, calib_rows=16)
from qdiff.block_recon import block_reconstruction
import edadm.recon as _er
_er.GRAPH_MIN_ITERS = 10 ** 9            # eager iterations: rocprofv3 then sees every launch (same kernels as the replayed graph)
g = torch.Generator().manual_seed(3)
N = 64
cali = (torch.randn(N, 3, 64, 64, generator=g).to(dev), torch.randint(1, 1000, (N,), generator=g).to(dev),
        torch.randn(N, 1, 512, generator=g).to(dev))
qnn.set_quant_state(True, True)
iters = int(os.environ.get("ITERS", "30"))
units = (("res 192@64", qnn.model.input_blocks[1][0]), ("tf 384@32", qnn.model.input_blocks[4][1].transformer_blocks[0]),
         ("res 960@8", qnn.model.middle_block[0]), ("tf 960@8", qnn.model.middle_block[1].transformer_blocks[0]),
         ("tf 576@16", qnn.model.input_blocks[7][1].transformer_blocks[0]),
         ("up 384->192@64", qnn.model.output_blocks[9][0]))
sel = os.environ.get("UNITS")            # comma-separated substrings of the names; default: the first three
units = [u for u in units if any(k in u[0] for k in sel.split(","))] if sel else units[:3]
for name, unit in units:
    kw = dict(cali_data=cali, iters=iters, act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-1, p=2.0,
              weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=32, input_prob=0.5, add_loss=0.8, recon_w=True,
              recon_a=True, keep_gpu=True)
    torch.cuda.synchronize(); t0 = time.time()
    block_reconstruction(qnn, unit, **kw)
    torch.cuda.synchronize()
    print(name, "total s", time.time() - t0, "iters", iters)
