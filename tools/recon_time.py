"""Diagnostic: steady-state ms per reconstruction iteration (HIP-graph replays, as the product runs them) of selected full-size
LDM-4 units.  UNITS = comma-separated name substrings; MIN_NK overrides contract.F16X3_MIN_NK."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev, calib_rows=16)
from qdiff.block_recon import block_reconstruction
import edadm.recon as er
from edadm import contract
if os.environ.get("DIRECT_SMALL"):
    contract.DIRECT_SMALL = os.environ["DIRECT_SMALL"] != "0"
if os.environ.get("WGRAD_SWAP"):
    contract.WGRAD_SWAP = os.environ["WGRAD_SWAP"] != "0"
if os.environ.get("SIDE_WGRAD"):
    contract.SIDE_WGRAD = os.environ["SIDE_WGRAD"] != "0"
if os.environ.get("INJECT"):
    er.INJECT_MODULE_LOSS = os.environ["INJECT"] != "0"
er.FP_FEAT_FORCE = True                     # short runs: the per-sample FP feature maps as the full-length job caches them
if os.environ.get("MIN_NK"):
    contract.F16X3_MIN_NK = int(os.environ["MIN_NK"])
g = torch.Generator().manual_seed(3)
N = 64
cali = (torch.randn(N, 3, 64, 64, generator=g).to(dev), torch.randint(1, 1000, (N,), generator=g).to(dev),
        torch.randn(N, 1, 512, generator=g).to(dev))
qnn.set_quant_state(True, True)
iters = int(os.environ.get("ITERS", "120"))
m = qnn.model
units = (("res 192@64", m.input_blocks[1][0]), ("tf 384@32", m.input_blocks[4][1].transformer_blocks[0]),
         ("res 384@32", m.input_blocks[5][0]), ("res 576@16", m.input_blocks[8][0]), ("tf 576@16", m.input_blocks[7][1].transformer_blocks[0]),
         ("res 960@8", m.middle_block[0]), ("tf 960@8", m.middle_block[1].transformer_blocks[0]),
         ("up 384->192@64", m.output_blocks[9][0]), ("up tf 384@32", m.output_blocks[6][1].transformer_blocks[0]))
sel = os.environ.get("UNITS")
units = [u for u in units if any(k in u[0] for k in sel.split(","))] if sel else units
for name, unit in units:
    kw = dict(cali_data=cali, iters=iters, act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-1, p=2.0,
              weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=32, input_prob=0.5, add_loss=0.8, recon_w=True,
              recon_a=True, keep_gpu=True)
    er.TIMING = {"iter_s": 0.0, "iters": 0}
    block_reconstruction(qnn, unit, **kw)
    print("%-16s %7.3f ms per iteration (%d steady iterations)" % (name, 1e3 * er.TIMING["iter_s"] / max(er.TIMING["iters"], 1), er.TIMING["iters"]))
er.TIMING = None
