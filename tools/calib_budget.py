"""Diagnostic: the full calibration job (bench.full_calibration) under other HBM budgets of the caches:  TRACE_GB / MEMO_GB / FEAT_GB."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "eda-dm_amd"))
import torch
import bench
from qdiff import data_utils as du
from edadm import recon as er
du.FP_TRACE_GB = float(os.environ.get("TRACE_GB", du.FP_TRACE_GB))
du.Q_MEMO_GB = float(os.environ.get("MEMO_GB", du.Q_MEMO_GB))
er.FP_FEAT_GB = float(os.environ.get("FEAT_GB", er.FP_FEAT_GB))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.full_calibration(dev)
print(json.dumps({k: r[k] for k in ("wall_s", "caching_s", "loop_s", "stages", "fp_trace", "fp_features", "peak_hbm_gb") if k in r}))
