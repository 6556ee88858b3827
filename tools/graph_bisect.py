"""Diagnostic: which units' reconstruction iteration survives HIP-graph capture (each case in its own process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["asblock:conv_in", "block:rb", "block:rb:nofeat", "block:at"]
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True)
        print("%-22s rc=%d %s" % (c, r.returncode, (r.stdout.strip().splitlines() or [""])[-1][:150]))
        if r.returncode and r.returncode > 0:
            print("    ", r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "")
    sys.exit(0)
for p in (ROOT, os.path.join(ROOT, "eda-dm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, random
case = sys.argv[1].split(":")
if "nofeat" in case:
    os.environ["EDADM_FP_FEAT_GB"] = "0"
if "f16x3off" in case:
    os.environ["EDADM_F16X3"] = "0"
if "threadlocal" in case:
    os.environ["EDADM_GRAPH_DEBUG"] = "threadlocal"
if "nohooks" in case:
    os.environ["EDADM_GRAPH_DEBUG"] = "nohooks"
for c in case:
    if c.startswith("dbg="):
        os.environ["EDADM_GRAPH_DEBUG"] = c[4:]
from helpers import build_toynet, WQ4, AQ8
from qdiff import QuantModel, set_weight_quantize_params, set_act_quantize_params
from qdiff.block_recon import block_reconstruction
from qdiff.layer_recon import layer_reconstruction
from qdiff.attn_layer_recon import AttnBlock_layer_reconstruction
import edadm.recon as recon
g = np.load(os.path.join(ROOT, "tests", "golden", "g8_recon.npz"))
aq = dict(AQ8); aq["prob"] = 1.0
qnn = QuantModel(build_toynet(g), WQ4, aq, sm_abit=8).cuda().eval()
x, t = torch.as_tensor(g["x"]).cuda(), torch.as_tensor(g["t"]).cuda()
set_weight_quantize_params(qnn, (x, t)); set_act_quantize_params(qnn, (x, t), batch_size=32)
kw = dict(cali_data=(x, t), iters=10, act_quant=True, asym=True, opt_mode="mse", lr_a=1e-4, lr_w=5e-2, p=2.0, weight=0.0001,
          b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=1.0, add_loss=0.0 if "noaddloss" in case else 0.8, recon_w="recon_w0" not in case, recon_a="recon_a0" not in case)
if "iters200" in case:
    kw["iters"] = 200
recon.GRAPH_MIN_ITERS = 4
if "noactq" in case:
    kw["act_quant"] = False
fn = {"layer": layer_reconstruction, "block": block_reconstruction, "asblock": block_reconstruction, "attn": AttnBlock_layer_reconstruction}[case[0]]
fn(qnn, getattr(qnn.model, case[1]), **kw)
torch.cuda.synchronize()
print("ok", case)
