"""What a rank of an N-rank calibration job sees in the data-parallel reconstruction iterations (edadm.recon.DP_LOOP): its
share of the 32-row minibatch is 32 / N rows.  No multi-GPU node is available to this build, but the per-rank kernels ARE measurable
on one GPU: steady-state ms per iteration (HIP-graph replays, as the product runs them) of the data-parallel-eligible unit classes of the
full-size LDM-4 at batch 32 / 16 / 8 / 4.  bench.py turns the ratios t(32 / N) / t(32) into `calibration.multi_rank.ceiling_measured_rows`
(in place of dividing the iteration time by N).       python tools/rank_rows.py out.json"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
qnn, sd, calib = bench.build_quantised_unet(dev, calib_rows=16)
from qdiff.block_recon import block_reconstruction
import edadm.recon as er
er.FP_FEAT_FORCE = True
g = torch.Generator().manual_seed(3)
N = 64
cali = (torch.randn(N, 3, 64, 64, generator=g).to(dev), torch.randint(1, 1000, (N,), generator=g).to(dev),
        torch.randn(N, 1, 512, generator=g).to(dev))
qnn.set_quant_state(True, True)
iters = int(os.environ.get("ITERS", "60"))
m = qnn.model
# one unit per class of the units with >= 1024 positions per row (DP_MIN_POSITIONS): (class key, positions per row, unit)
units = (("res@4096", 4096, m.input_blocks[1][0]), ("res@1024", 1024, m.input_blocks[5][0]),
         ("tf@1024", 1024, m.input_blocks[4][1].transformer_blocks[0]), ("up@4096", 4096, m.output_blocks[9][0]))
out = {"iters": iters, "note": "ms per steady-state reconstruction iteration at the shipped hyper-parameters; rows = minibatch of the iteration",
       "classes": {}}
for name, pos, unit in units:
    row = {}
    for bs in (32, 16, 8, 4):
        kw = dict(cali_data=cali, iters=iters, act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-1, p=2.0,
                  weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=bs, input_prob=0.5, add_loss=0.8, recon_w=True,
                  recon_a=True, keep_gpu=True)
        er.TIMING = {"iter_s": 0.0, "iters": 0}
        block_reconstruction(qnn, unit, **kw)
        row[str(bs)] = 1e3 * er.TIMING["iter_s"] / max(er.TIMING["iters"], 1)
    er.TIMING = None
    out["classes"][name] = {"positions_per_row": pos, "ms_per_iteration": row, "ratio_to_32_rows": {k: v / row["32"] for k, v in row.items()}}
    print("%-10s" % name, "  ".join("%2s rows %.3f ms (x%.2f)" % (k, v, v / row["32"]) for k, v in row.items()), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
