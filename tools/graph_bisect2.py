"""Diagnostic: HIP-graph capture of pieces of the training graph, each in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["gn_fwd", "gn_fwdbwd", "gn_silu_fwdbwd", "silu_fwdbwd", "conv_fwdbwd", "rb_fwd", "rb_fwdbwd", "rb_fwdbwd_fp", "at_fwd", "at_fwdbwd",
         "softmax_fwdbwd", "bmm_fwdbwd", "transpose_fwdbwd", "add_fwdbwd", "linear_fwdbwd", "rb_parts"]
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True)
        print("%-18s rc=%d %s" % (c, r.returncode, (r.stdout.strip().splitlines() or [""])[-1][:150]))
    sys.exit(0)
for p in (ROOT, os.path.join(ROOT, "eda-dm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from helpers import build_toynet, WQ4, AQ8
from qdiff import QuantModel, set_weight_quantize_params, set_act_quantize_params
from edadm import train_ops as T
import torch.nn as nn
case = sys.argv[1]
g = np.load(os.path.join(ROOT, "tests", "golden", "g8_recon.npz"))
aq = dict(AQ8); aq["prob"] = 1.0
qnn = QuantModel(build_toynet(g), WQ4, aq, sm_abit=8).cuda().eval()
x, t = torch.as_tensor(g["x"]).cuda()[:16], torch.as_tensor(g["t"]).cuda()[:16]
set_weight_quantize_params(qnn, (x, t)); set_act_quantize_params(qnn, (x, t), batch_size=16)
h = torch.randn(16, 32, 8, 8, device="cuda", requires_grad=True)
temb = torch.randn(16, 64, device="cuda")
norm = nn.GroupNorm(32, 32).cuda()
rb, at = qnn.model.rb, qnn.model.at


def fn():
    if case == "gn_fwd":
        with torch.no_grad():
            return T.group_norm(h, norm)
    if case == "gn_fwdbwd":
        T.group_norm(h, norm).sum().backward()
    if case == "gn_silu_fwdbwd":
        T.group_norm(h, norm, silu=True).sum().backward()
    if case == "silu_fwdbwd":
        T.silu(h).sum().backward()
    if case == "conv_fwdbwd":
        rb.conv1(h).sum().backward()
    if case == "linear_fwdbwd":
        tt = temb.clone().requires_grad_(True)
        rb.temb_proj(tt).sum().backward()
    if case == "rb_fwd":
        with torch.no_grad():
            return rb(h, temb)
    if case == "rb_fwdbwd":
        rb(h, temb).sum().backward()
    if case == "rb_fwdbwd_fp":
        rb.set_quant_state(False, False)
        rb(h, temb).sum().backward()
    if case == "rb_parts":
        a = rb.norm1(h, silu=True)
        b = rb.conv1(a)
        c = b + rb.temb_proj(T.silu(temb))[:, :, None, None]
        c.sum().backward()
    if case == "at_fwd":
        with torch.no_grad():
            return at(h)
    if case == "at_fwdbwd":
        at(h).sum().backward()
    if case == "softmax_fwdbwd":
        s = torch.randn(16, 64, 64, device="cuda", requires_grad=True)
        T.softmax(s).sum().backward()
    if case == "bmm_fwdbwd":
        a = torch.randn(16, 64, 32, device="cuda", requires_grad=True)
        b = torch.randn(16, 64, 32, device="cuda", requires_grad=True)
        T.bmm_nt(a, b, 0.5).sum().backward()
    if case == "transpose_fwdbwd":
        a = torch.randn(16, 64, 32, device="cuda", requires_grad=True)
        T.transpose12(a).sum().backward()
    if case == "add_fwdbwd":
        (h + temb[:, :32, None, None]).sum().backward()


rb.set_quant_state(True, True); at.set_quant_state(True, True)
for _ in range(2):
    fn()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    fn()
gr.replay()
torch.cuda.synchronize()
print("ok", case)
