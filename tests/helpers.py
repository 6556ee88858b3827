"""Test helpers: rebuild the tiny fixture networks in the product's module classes."""
from types import SimpleNamespace

import numpy as np
import torch

WQ4 = dict(n_bits=4, symmetric=True, channel_wise=True, scale_method="mse")
AQ8 = dict(n_bits=8, symmetric=True, channel_wise=False, scale_method="mse", leaf_param=True, prob=0.5)


def sub_sd(g, prefix):
    return {k[len(prefix):]: torch.as_tensor(np.asarray(g[k])) for k in g.files if k.startswith(prefix)}


def build_cifar(g):
    from edadm.nets.ddpm_unet import Model
    cfg = SimpleNamespace(
        model=SimpleNamespace(type="simple", in_channels=3, out_ch=3, ch=int(g["cfg/ch"]),
                              ch_mult=[int(v) for v in g["cfg/ch_mult"]], num_res_blocks=int(g["cfg/nres"]),
                              attn_resolutions=[int(v) for v in g["cfg/attn"]], dropout=0.0, resamp_with_conv=True),
        data=SimpleNamespace(image_size=int(g["cfg/res"])),
        diffusion=SimpleNamespace(num_diffusion_timesteps=1000))
    m = Model(cfg)
    m.load_state_dict(sub_sd(g, "sd/"))
    return m.eval()


def build_ldm(g):
    from edadm.nets.ldm_unet import UNetModel
    kw = {}
    for k in g.files:
        if k.startswith("cfg/"):
            v = g[k]
            kw[k[4:]] = v.tolist() if v.ndim else v.item()
    m = UNetModel(**kw)
    m.load_state_dict(sub_sd(g, "sd/"))
    return m.eval()


def ldm_state_dict_shapes(g):
    """(key, shape) of every state_dict entry of the fixture's UNetModel config, in the product's own classes (the
    names and shapes are those of the reference's, which is what makes its fixtures loadable)."""
    from edadm.nets.ldm_unet import UNetModel
    kw = {}
    for k in g.files:
        if k.startswith("cfg/"):
            v = g[k]
            kw[k[4:]] = v.tolist() if v.ndim else v.item()
    with torch.device("meta"):
        m = UNetModel(**kw)
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]


def build_ldm_formula(g):
    """UNetModel of a fixture whose weights are formula weights (tests/golden/_weights.py), not stored."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _weights import formula_state_dict
    from edadm.nets.ldm_unet import UNetModel
    kw = {}
    for k in g.files:
        if k.startswith("cfg/"):
            v = g[k]
            kw[k[4:]] = v.tolist() if v.ndim else v.item()
    m = UNetModel(**kw)
    sd = formula_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], int(g["weights_seed"]))
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    return m.eval()


def quantize_like_reference(model, g, kind, split=True):
    """QuantModel wrapped and configured exactly as the fixture generator did, with the
    reference's own deltas / zero points loaded."""
    from qdiff import QuantModel
    from edadm.state import load_quant_state
    qnn = QuantModel(model, WQ4, AQ8, sm_abit=8)
    qnn.cuda().eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    if kind == "cifar":
        qnn.model.config.split_shortcut = split
    else:
        qnn.set_grad_ckpt(False)
        qnn.model.split_shortcut = split
    x = torch.as_tensor(g["x"]).cuda()
    t = torch.as_tensor(g["t"]).cuda()
    ctx = torch.as_tensor(g["ctx"]).cuda() if "ctx" in g.files else None
    with torch.no_grad():
        out_fp = qnn(x[:2], t[:2], None if ctx is None else ctx[:2])     # FP pass: creates the split quantizers
    qnn.cuda()
    n = load_quant_state(qnn, {k: g[k] for k in g.files if k.startswith("qp/")}, prefix="qp/")
    return qnn, (x, t, ctx), n


def build_toynet(g):
    """The fixture's 2-block toy model (tests/golden/make_golden.py::_ToyNet) in product classes."""
    import torch.nn as nn
    from edadm.nets.ddpm_unet import ResnetBlock, AttnBlock

    class ToyNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.in_channels = 3
            self.conv_in = nn.Conv2d(3, 32, 3, padding=1)
            self.temb_lin = nn.Linear(8, 64)
            self.rb = ResnetBlock(in_channels=32, out_channels=32, dropout=0.0, temb_channels=64)
            self.at = AttnBlock(32)
            self.conv_out = nn.Conv2d(32, 3, 3, padding=1)

        def forward(self, x, t, context=None):
            temb = self.temb_lin(torch.stack([torch.sin(t * (i + 1) * 0.01) for i in range(8)], 1))
            return self.conv_out(self.at(self.rb(self.conv_in(x), temb)))

    m = ToyNet()
    m.load_state_dict(sub_sd(g, "sd/"))
    return m.eval()
