"""Generate golden input/output vectors by IMPORTING the reference (read-only at
/root/reference) on CPU in the build container.  Run:  python tests/golden/make_golden.py

Only data leaves this script: inputs and the reference's outputs, written as small .npz
fixtures next to it.  The reference's source never travels.  Every fixture
records the reference file:line that produced it in the README table of tests/golden/.
"""
import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

_refshim.install()

from qdiff.quant_layer import UniformAffineQuantizer, QuantModule, lp_loss  # noqa: E402
from qdiff.adaptive_rounding import AdaRoundQuantizer  # noqa: E402
from qdiff.block_recon import LinearTempDecay, LossFunction, block_reconstruction  # noqa: E402
from qdiff.layer_recon import layer_reconstruction  # noqa: E402
from qdiff.quant_model import QuantModel  # noqa: E402
from qdiff.quant_block import (QuantResnetBlock, QuantAttnBlock, QuantResBlock,  # noqa: E402
                               QuantBasicTransformerBlock, BaseQuantBlock, QuantAttentionBlock)
from qdiff.set_quantize_params import set_weight_quantize_params, set_act_quantize_params  # noqa: E402
from qdiff.data_utils import save_inp_oup_data  # noqa: E402
from qdiff.recon_block_Qmodel import recon_block_Qmodel, Change_LDM_model_attnblock  # noqa: E402
from qdiff.utils import seed_everything  # noqa: E402
from ddim.models.diffusion import Model as DDPMModel, ResnetBlock, AttnBlock  # noqa: E402
from ddim.functions.denoising import compute_alpha, generalized_steps  # noqa: E402
from ldm.modules.diffusionmodules.openaimodel import UNetModel, ResBlock, AttentionBlock  # noqa: E402
from ldm.modules.attention import BasicTransformerBlock  # noqa: E402
from ldm.modules.diffusionmodules.util import (make_ddim_timesteps, make_ddim_sampling_parameters,  # noqa: E402
                                               make_beta_schedule, timestep_embedding)


def A(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, d):
    flat = {}
    for k, v in d.items():
        flat[k] = A(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def reinit_zero_modules(model, gen):
    """zero_module() convs are all-zero at init (openaimodel.py:229-231,315,721); give them
    small random values so the fixtures exercise them."""
    for p in model.parameters():
        if p.numel() > 0 and float(p.detach().abs().max()) == 0.0:
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)


def qparams_of(qnn):
    """Collect (delta, zero_point, n_bits) of every quantizer, keyed by module path."""
    out = {}
    for name, m in qnn.named_modules():
        if isinstance(m, (UniformAffineQuantizer, AdaRoundQuantizer)):
            if m.delta is None:
                continue
            out["qp/%s/delta" % name] = A(m.delta).astype(np.float32)
            out["qp/%s/zero_point" % name] = A(m.zero_point).astype(np.float32)
            out["qp/%s/n_bits" % name] = np.int64(m.n_bits)
    return out


# --------------------------------------------------------------------------------------
def g1_weight_init():
    """G1: weight UAQ init, per-channel (quant_layer.py:95-105,120-147,150-213,215-226)."""
    g = torch.Generator().manual_seed(101)
    d = {}
    cases = []
    w = torch.randn(8, 6, 3, 3, generator=g) * 0.2
    cases.append(("conv_two", w))
    cases.append(("lin_two", torch.randn(12, 20, generator=g) * 0.5))
    cases.append(("conv_pos", (torch.rand(8, 4, 3, 3, generator=g) * 0.3)))
    cases.append(("conv_neg", -(torch.rand(8, 4, 1, 1, generator=g) * 0.7)))
    wz = torch.randn(6, 5, 3, 3, generator=g) * 0.1
    wz[2] = 0.0  # an all-zero channel -> eps clamp path
    cases.append(("conv_zero_ch", wz))
    for cname, w in cases:
        d["w/%s" % cname] = w
        for bits in (4, 8):
            for sym in (True, False):
                if not sym and bits == 8 and cname not in ("conv_pos", "conv_neg"):
                    # 2-D search at 8 bit is 100x256 candidates: keep one small case only
                    if cname != "lin_two":
                        continue
                q = UniformAffineQuantizer(n_bits=bits, symmetric=sym, channel_wise=True, scale_method="mse")
                out = q(w)
                key = "%s/b%d/%s" % (cname, bits, "sym" if sym else "asym")
                d[key + "/delta"] = q.delta
                d[key + "/zero_point"] = q.zero_point
                d[key + "/one_side"] = np.array({"pos": 1, "neg": -1, "no": 0}[q.one_side_dist])
                d[key + "/out"] = out
    save("g1_weight_init", d)


def g1b_max_init():
    """G1b: scale_method='max' -- the constructor DEFAULT (quant_layer.py:48) -- init_quantization_scale_2 (:278-330): per
    channel for weights, per tensor for activations (leaf_param / always_zero variants), and the forward that follows.
    Note the reference's symmetric 'max' rule: delta = absmax / n_levels with zero_point 0, so negatives clamp to code 0."""
    g = torch.Generator().manual_seed(111)
    d = {}
    wz = torch.randn(6, 5, 3, 3, generator=g) * 0.1
    wz[2] = 0.0
    cases = [("conv_two", torch.randn(8, 6, 3, 3, generator=g) * 0.2), ("lin_two", torch.randn(12, 20, generator=g) * 0.5),
             ("conv1d_two", torch.randn(6, 10, 1, generator=g) * 0.3), ("conv_pos", torch.rand(8, 4, 3, 3, generator=g) * 0.3),
             ("conv_neg", -(torch.rand(8, 4, 1, 1, generator=g) * 0.7)), ("conv_zero_ch", wz)]
    import warnings
    for cname, w in cases:
        d["w/%s" % cname] = w
        for bits in (4, 8):
            for sym in (True, False):
                q = UniformAffineQuantizer(n_bits=bits, symmetric=sym, channel_wise=True, scale_method="max")
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out = q(w)
                key = "%s/b%d/%s" % (cname, bits, "sym" if sym else "asym")
                d[key + "/delta"], d[key + "/zero_point"], d[key + "/out"] = q.delta, q.zero_point, out
    acts = [("act_two", torch.randn(4, 16, 8, 8, generator=g) * 1.7), ("act_pos", torch.rand(4, 16, 8, 8, generator=g) * 3.0),
            ("act_tokens", torch.randn(2, 33, 24, generator=g))]
    for aname, x in acts:
        d["x/%s" % aname] = x
        for sym in (True, False):
            for az in (False, True):
                q = UniformAffineQuantizer(n_bits=8, symmetric=sym, channel_wise=False, scale_method="max", leaf_param=True,
                                           always_zero=az)
                out = q(x)
                key = "%s/%s/%s" % (aname, "sym" if sym else "asym", "az" if az else "nz")
                d[key + "/delta"], d[key + "/zero_point"], d[key + "/out"] = q.delta, np.float32(q.zero_point), out
                assert isinstance(q.delta, nn.Parameter)
    save("g1b_max_init", d)


def g2_act_init():
    """G2: act UAQ init with EMA over batches (quant_layer.py:79-85,150-199,246-264)."""
    g = torch.Generator().manual_seed(202)
    d = {}
    for cname, gen in (("two", lambda: torch.randn(4, 8, 6, 6, generator=g) * 1.5),
                       ("pos", lambda: torch.rand(4, 8, 6, 6, generator=g) * 2.0),
                       ("softmax", lambda: torch.softmax(torch.randn(4, 16, 16, generator=g) * 2, -1))):
        for bits in (8, 4):
            for sym in (True, False):
                if not sym and bits == 8:
                    continue  # per-tensor 2-D search at 8 bit: 25600 passes, skip
                q = UniformAffineQuantizer(n_bits=bits, symmetric=sym, channel_wise=False, scale_method="mse",
                                           leaf_param=True, prob=0.5)
                g.manual_seed(202 + bits)
                for k in range(4):
                    x = gen()
                    out = q(x)
                    key = "%s/b%d/%s/step%d" % (cname, bits, "sym" if sym else "asym", k)
                    d[key + "/x"] = x
                    d[key + "/delta"] = q.delta
                    d[key + "/zero_point"] = q.zero_point
                    d[key + "/running_min"] = q.running_min
                    d[key + "/running_max"] = q.running_max
                    d[key + "/out"] = out
    save("g2_act_init", d)


def g3_uaq_forward():
    """G3: inited UAQ forward/backward incl. injected prob mask (quant_layer.py:266-276)."""
    g = torch.Generator().manual_seed(303)
    d = {}
    x = (torch.randn(4, 8, 6, 6, generator=g) * 1.3)
    gy = torch.randn(4, 8, 6, 6, generator=g)
    mask_u = torch.rand(4, 8, 6, 6, generator=g)
    d["x"], d["gy"], d["mask_u"] = x, gy, mask_u
    for bits, delta, zp in ((8, 0.021, 128.0), (8, 0.013, 127.0), (4, 0.31, 7.0), (8, 0.009, 0.0)):
        q = UniformAffineQuantizer(n_bits=bits, symmetric=True, channel_wise=False, scale_method="mse",
                                   leaf_param=True, prob=0.5)
        q.delta = nn.Parameter(torch.tensor(delta))
        q.zero_point = torch.tensor(zp)
        q.inited = True
        key = "b%d_d%g_z%g" % (bits, delta, zp)
        xr = x.clone().requires_grad_(True)
        out = q(xr)
        (out * gy).sum().backward()
        codes = torch.clamp(torch.round(x / delta) + zp, 0, 2 ** bits - 1)
        d[key + "/out"], d[key + "/codes"] = out, codes
        d[key + "/gx"], d[key + "/gdelta"] = xr.grad, q.delta.grad
        # training mode with the mask injected through a patched rand_like
        q.delta.grad = None
        q.is_training = True
        orig = torch.rand_like
        torch.rand_like = lambda t, **k: mask_u
        try:
            xr = x.clone().requires_grad_(True)
            out = q(xr)
            (out * gy).sum().backward()
        finally:
            torch.rand_like = orig
        d[key + "/train_out"], d[key + "/train_gx"], d[key + "/train_gdelta"] = out, xr.grad, q.delta.grad
    save("g3_uaq_forward", d)


def g4_adaround():
    """G4: AdaRound alpha init / soft / hard / grad (adaptive_rounding.py:39-72)."""
    g = torch.Generator().manual_seed(404)
    d = {}
    for cname, w, bits in (("conv", torch.randn(8, 6, 3, 3, generator=g) * 0.2, 4),
                           ("lin", torch.randn(10, 16, generator=g) * 0.4, 8)):
        uaq = UniformAffineQuantizer(n_bits=bits, symmetric=True, channel_wise=True, scale_method="mse")
        uaq(w)
        aq = AdaRoundQuantizer(uaq=uaq, round_mode="learned_hard_sigmoid", weight_tensor=w.clone())
        d[cname + "/w"], d[cname + "/delta"], d[cname + "/zero_point"] = w, aq.delta, aq.zero_point
        d[cname + "/alpha0"] = aq.alpha.detach().clone()
        aq.soft_targets = True
        gy = torch.randn(w.shape, generator=g)
        out = aq(w)
        (out * gy).sum().backward()
        d[cname + "/gy"], d[cname + "/soft_out"], d[cname + "/galpha"] = gy, out, aq.alpha.grad.clone()
        # perturb alpha so hard rounding differs from nearest
        with torch.no_grad():
            aq.alpha.add_(torch.randn(w.shape, generator=g) * 2.0)
        d[cname + "/alpha1"] = aq.alpha
        aq.alpha.grad = None
        out = aq(w)
        (out * gy).sum().backward()
        d[cname + "/soft_out1"], d[cname + "/galpha1"] = out, aq.alpha.grad
        aq.soft_targets = False
        d[cname + "/hard_out1"] = aq(w)
    save("g4_adaround", d)


def g5_loss():
    """G5: lp_loss fwd/bwd, LinearTempDecay, dead round-loss branch (quant_layer.py:26-33,
    block_recon.py:235-323)."""
    g = torch.Generator().manual_seed(505)
    d = {}
    for cname, shape in (("4d", (4, 8, 6, 6)), ("2d", (4, 24)), ("3d", (4, 9, 16))):
        p = torch.randn(shape, generator=g).requires_grad_(True)
        t = torch.randn(shape, generator=g)
        l = lp_loss(p, t, p=2.0)
        l.backward()
        d[cname + "/pred"], d[cname + "/tgt"], d[cname + "/loss"], d[cname + "/gpred"] = p, t, l, p.grad
        d[cname + "/loss_all"] = lp_loss(p, t, p=2.4, reduction="all")
    td = LinearTempDecay(100, rel_start_decay=0.2, start_b=20, end_b=2)
    d["temp/t"] = np.arange(0, 101)
    d["temp/b"] = np.array([td(t) for t in range(0, 101)], dtype=np.float64)
    save("g5_loss", d)


def _mk_qmodule(org, wbits=4, abits=8):
    wq = {"n_bits": wbits, "symmetric": True, "channel_wise": True, "scale_method": "mse"}
    aq = {"n_bits": abits, "symmetric": True, "channel_wise": False, "scale_method": "mse",
          "leaf_param": True, "prob": 0.5}
    return QuantModule(org, wq, aq), wq, aq


def g6_quant_module():
    """G6: QuantModule forward variants (quant_layer.py:406-437)."""
    torch.manual_seed(606)
    g = torch.Generator().manual_seed(606)
    d = {}
    cases = [
        ("conv3", nn.Conv2d(32, 48, 3, padding=1), torch.randn(4, 32, 8, 8, generator=g), 0, None),
        ("conv3s2", nn.Conv2d(32, 32, 3, stride=2, padding=0),
         torch.nn.functional.pad(torch.randn(4, 32, 8, 8, generator=g), (0, 1, 0, 1)), 0, None),
        ("conv1split", nn.Conv2d(64, 32, 1), torch.randn(4, 64, 8, 8, generator=g) *
         torch.cat([torch.ones(32), 3 * torch.ones(32)]).view(1, 64, 1, 1), 32, None),
        ("conv1d", nn.Conv1d(32, 96, 1), torch.randn(4, 32, 16, generator=g), 0, None),
        ("linear", nn.Linear(32, 64), torch.randn(4, 32, generator=g), 0, None),
        ("linear3d", nn.Linear(32, 64, bias=False), torch.randn(4, 16, 32, generator=g), 0, None),
    ]
    for cname, org, x, split, _ in cases:
        qm, _, _ = _mk_qmodule(org)
        d[cname + "/weight"] = org.weight
        if org.bias is not None:
            d[cname + "/bias"] = org.bias
        d[cname + "/x"] = x
        d[cname + "/out_fp"] = qm(x, split) if split else qm(x)
        qm.set_quant_state(True, False)
        d[cname + "/out_w"] = qm(x)
        qm.weight_quantizer.set_inited(True)
        if split:
            qm.weight_quantizer_0.set_inited(True)
        qm.set_quant_state(True, True)
        d[cname + "/out_wa"] = qm(x)
        qm.act_quantizer.set_inited(True)
        d[cname + "/w_delta"], d[cname + "/w_zp"] = qm.weight_quantizer.delta, qm.weight_quantizer.zero_point
        d[cname + "/a_delta"], d[cname + "/a_zp"] = qm.act_quantizer.delta, qm.act_quantizer.zero_point
        if split:
            qm.act_quantizer_0.set_inited(True)
            d[cname + "/w_delta_0"], d[cname + "/w_zp_0"] = qm.weight_quantizer_0.delta, qm.weight_quantizer_0.zero_point
            d[cname + "/a_delta_0"], d[cname + "/a_zp_0"] = qm.act_quantizer_0.delta, qm.act_quantizer_0.zero_point
        d[cname + "/split"] = np.int64(split)
    save("g6_quant_module", d)


# --------------------------------------------------------------------------------------
def cifar_cfg(ch=32, ch_mult=(1, 2, 2), nres=2, attn=(8,), res=16):
    return SimpleNamespace(
        model=SimpleNamespace(type="simple", in_channels=3, out_ch=3, ch=ch, ch_mult=list(ch_mult),
                              num_res_blocks=nres, attn_resolutions=list(attn), dropout=0.0,
                              resamp_with_conv=True),
        data=SimpleNamespace(image_size=res),
        diffusion=SimpleNamespace(num_diffusion_timesteps=1000))


WQ4 = {"n_bits": 4, "symmetric": True, "channel_wise": True, "scale_method": "mse"}
AQ8 = {"n_bits": 8, "symmetric": True, "channel_wise": False, "scale_method": "mse", "leaf_param": True,
       "prob": 0.5}


def sd_of(model):
    return {"sd/" + k: v for k, v in model.state_dict().items()}


def g13_cifar_unet(wbits=4, fname="g13_cifar_unet", seed=1301):
    """G13a: tiny DDPM UNet (ddim/models/diffusion.py:199-392) FP and fake-quant forward after
    the reference's own scale-init drivers (set_quantize_params.py:9-71), split shortcut on.
    wbits=8 -> g13_cifar_w8: BASELINE config 1's bit widths (W8A8, sample_diffusion_ddim.py --weight_bit 8)."""
    seed_everything(seed)
    g = torch.Generator().manual_seed(seed)
    cfg = cifar_cfg()
    model = DDPMModel(cfg).eval()
    d = sd_of(model)
    d["cfg/ch"], d["cfg/ch_mult"], d["cfg/nres"], d["cfg/attn"], d["cfg/res"] = 32, [1, 2, 2], 2, [8], 16
    x = torch.randn(8, 3, 16, 16, generator=g)
    t = torch.tensor([999., 870., 640., 500., 333., 120., 40., 0.])
    d["x"], d["t"] = x, t
    with torch.no_grad():
        d["out_fp"] = model(x, t)
    if fname != "g13_cifar_unet":
        # same seed -> same weights and inputs as g13_cifar_unet: the variant stores only what differs
        base = np.load(os.path.join(HERE, "g13_cifar_unet.npz"))
        assert all(np.array_equal(base[k], A(v)) for k, v in d.items() if k.startswith("sd/"))
        d = {k: v for k, v in d.items() if not k.startswith("sd/")}
    wq = dict(WQ4)
    wq["n_bits"] = wbits
    d["cfg/wbits"] = wbits
    qnn = QuantModel(model, wq, AQ8, sm_abit=8)
    qnn.eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.model.config.split_shortcut = True
    cali = (x, t)
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali, batch_size=4)   # 2 batches -> EMA exercised
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        d["out_q"] = qnn(x, t)
        qnn.set_quant_state(True, False)
        d["out_wq"] = qnn(x, t)
    d.update(qparams_of(qnn))
    # G11 structure: unit order as the reference walks it
    units = []

    class Spy(recon_block_Qmodel):
        pass

    rb = sys.modules['qdiff.recon_block_Qmodel']
    rec = []
    ob, ol = rb.block_reconstruction, rb.layer_reconstruction
    rb.block_reconstruction = lambda m, blk, **k: rec.append(("block", blk))
    rb.layer_reconstruction = lambda m, lay, **k: rec.append(("layer", lay))
    try:
        qnn.set_quant_state(True, True)
        recon_block_Qmodel(None, qnn, cali, {}).recon()
    finally:
        rb.block_reconstruction, rb.layer_reconstruction = ob, ol
    names = {m: n for n, m in qnn.named_modules()}
    d["units"] = np.array(["%s:%s:%s" % (k, names[m], type(m).__name__) for k, m in rec])
    # the --layer_recon walk (recon_layer_Qmodel.py:20-120) on the same model: unit order only
    import qdiff.recon_layer_Qmodel  # noqa: F401
    rl = sys.modules['qdiff.recon_layer_Qmodel']
    lrec = []
    ol, oa = rl.layer_reconstruction, rl.AttnBlock_layer_reconstruction
    rl.layer_reconstruction = lambda m, lay, **k: lrec.append(("layer", lay))
    rl.AttnBlock_layer_reconstruction = lambda m, blk, **k: lrec.append(("attn", blk))
    try:
        rl.recon_layer_Qmodel(None, qnn, cali, {}).recon()
    finally:
        rl.layer_reconstruction, rl.AttnBlock_layer_reconstruction = ol, oa
    d["layer_units"] = np.array(["%s:%s" % (k, names[m]) for k, m in lrec])
    save(fname, d)
    return model, qnn, cali


def ldm_kwargs(kind):
    if kind == "imagenet":   # cin256-v2.yaml shape family, shrunk
        return dict(image_size=8, in_channels=3, out_channels=3, model_channels=32,
                    attention_resolutions=[2, 1], num_res_blocks=1, channel_mult=[1, 2],
                    num_heads=1, use_spatial_transformer=True, transformer_depth=1, context_dim=16)
    if kind == "church":     # lsun_churches256 shape family, shrunk
        return dict(image_size=8, in_channels=4, out_channels=4, model_channels=32,
                    attention_resolutions=[2, 1], num_res_blocks=1, channel_mult=[1, 2],
                    num_heads=2, use_scale_shift_norm=True, resblock_updown=True)
    raise ValueError(kind)


def g13_ldm_unet(kind):
    """G13b/c: tiny LDM UNetModel (openaimodel.py:447-783) FP and fake-quant forward."""
    seed_everything(1302)
    g = torch.Generator().manual_seed(1302 + len(kind))
    kw = ldm_kwargs(kind)
    model = UNetModel(**kw).eval()
    reinit_zero_modules(model, g)
    d = sd_of(model)
    for k, v in kw.items():
        d["cfg/" + k] = np.asarray(v)
    cin = kw["in_channels"]
    x = torch.randn(8, cin, 8, 8, generator=g)
    t = torch.tensor([981, 800, 640, 500, 333, 120, 40, 1]).long()
    ctx = torch.randn(8, 1, 16, generator=g) if kind == "imagenet" else None
    d["x"], d["t"] = x, t
    if ctx is not None:
        d["ctx"] = ctx
    with torch.no_grad():
        d["out_fp"] = model(x, t, ctx)
    aq = dict(AQ8)
    qnn = QuantModel(model, WQ4, aq, sm_abit=8, act_quant_mode="qdiff")
    qnn.eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_grad_ckpt(False)
    qnn.model.split_shortcut = True
    cali = (x, t, ctx) if ctx is not None else (x, t)
    set_weight_quantize_params(qnn, cali)
    # the generic driver only flips QuantModule/QuantAttnBlock quantizers (set_quantize_params.py:17-27);
    # the LDM/conditional drivers also flip q/k/v/w of the attention wrappers
    # (set_quantize_params_Conditional.py:31-46).  Quantizers start with inited=False, so one
    # pass initialises every one of them either way.
    set_act_quantize_params(qnn, cali, batch_size=4)
    for m in qnn.modules():
        if isinstance(m, UniformAffineQuantizer):
            m.set_inited(True)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        d["out_q"] = qnn(x, t, ctx)
        qnn.set_quant_state(True, False)
        d["out_wq"] = qnn(x, t, ctx)
    d.update(qparams_of(qnn))
    rb = sys.modules['qdiff.recon_block_Qmodel']
    rec = []
    ob, ol = rb.block_reconstruction, rb.layer_reconstruction
    rb.block_reconstruction = lambda m, blk, **k: rec.append(("block", blk))
    rb.layer_reconstruction = lambda m, lay, **k: rec.append(("layer", lay))
    try:
        recon_block_Qmodel(None, qnn, cali, {}).recon()
    finally:
        rb.block_reconstruction, rb.layer_reconstruction = ob, ol
    names = {m: n for n, m in qnn.named_modules()}
    d["units"] = np.array(["%s:%s:%s" % (k, names[m], type(m).__name__) for k, m in rec])
    save("g13_ldm_%s" % kind, d)


def g7_blocks():
    """G7: block forwards FP vs fake-quant with the reference's own initialised qparams
    (quant_block.py:46-116,168-192,204-235,238-285,300-348,398-451)."""
    seed_everything(707)
    g = torch.Generator().manual_seed(707)
    d = {}
    # --- CIFAR blocks
    rb = ResnetBlock(in_channels=64, out_channels=32, dropout=0.0, temb_channels=64).eval()
    at = AttnBlock(32).eval()
    holder = nn.Module()
    holder.in_channels = 3
    holder.rb, holder.at = rb, at
    for k, v in holder.state_dict().items():
        d["cifar/sd/" + k] = v
    qh = QuantModel(holder, WQ4, AQ8, sm_abit=8)
    x = torch.randn(4, 64, 8, 8, generator=g) * torch.cat([torch.ones(32), 2.5 * torch.ones(32)]).view(1, 64, 1, 1)
    temb = torch.randn(4, 64, generator=g)
    xa = torch.randn(4, 32, 8, 8, generator=g)
    d["cifar/x"], d["cifar/temb"], d["cifar/xa"] = x, temb, xa
    qrb, qat = qh.model.rb, qh.model.at
    assert isinstance(qrb, QuantResnetBlock) and isinstance(qat, QuantAttnBlock)
    with torch.no_grad():
        d["cifar/rb_fp"] = qrb(x, temb, split=32)
        d["cifar/at_fp"] = qat(xa)
        qh.set_quant_state(True, True)
        d["cifar/rb_q0"] = qrb(x, temb, split=32)   # first call initialises every quantizer
        d["cifar/at_q0"] = qat(xa)
        for m in qh.modules():
            if isinstance(m, UniformAffineQuantizer):
                m.set_inited(True)
        d["cifar/rb_q"] = qrb(x, temb, split=32)
        d["cifar/at_q"] = qat(xa)
    for k, v in qparams_of(qh).items():
        d["cifar/" + k] = v
    # --- LDM blocks
    res = ResBlock(32, 64, 0.0, out_channels=64).eval()
    res_ss = ResBlock(32, 64, 0.0, out_channels=32, use_scale_shift_norm=True, down=True).eval()
    res_up = ResBlock(32, 64, 0.0, out_channels=32, up=True).eval()
    tr = BasicTransformerBlock(32, 2, 16, context_dim=24, checkpoint=False).eval()
    ab = AttentionBlock(32, num_heads=2).eval()
    holder = nn.Module()
    holder.in_channels = 3
    holder.res, holder.res_ss, holder.res_up, holder.tr, holder.ab = res, res_ss, res_up, tr, ab
    reinit_zero_modules(holder, g)
    for k, v in holder.state_dict().items():
        d["ldm/sd/" + k] = v
    qh = QuantModel(holder, WQ4, AQ8, sm_abit=8)
    x = torch.randn(4, 32, 8, 8, generator=g)
    emb = torch.randn(4, 64, generator=g)
    xs = torch.randn(4, 16, 32, generator=g)
    ctx1 = torch.randn(4, 1, 24, generator=g)
    ctx7 = torch.randn(4, 7, 24, generator=g)
    d["ldm/x"], d["ldm/emb"], d["ldm/xs"], d["ldm/ctx1"], d["ldm/ctx7"] = x, emb, xs, ctx1, ctx7
    m = qh.model
    assert isinstance(m.res, QuantResBlock) and isinstance(m.tr, QuantBasicTransformerBlock)
    with torch.no_grad():
        d["ldm/res_fp"] = m.res(x, emb)
        d["ldm/res_ss_fp"] = m.res_ss(x, emb)
        d["ldm/res_up_fp"] = m.res_up(x, emb)
        d["ldm/tr_fp7"] = m.tr(xs, ctx7)
        d["ldm/tr_fp1"] = m.tr(xs, ctx1)
        d["ldm/ab_fp"] = m.ab(x)
        qh.set_quant_state(True, True)
        m.res(x, emb), m.res_ss(x, emb), m.res_up(x, emb), m.tr(xs, ctx7), m.ab(x)
        for mm in qh.modules():
            if isinstance(mm, UniformAffineQuantizer):
                mm.set_inited(True)
        d["ldm/res_q"] = m.res(x, emb)
        d["ldm/res_ss_q"] = m.res_ss(x, emb)
        d["ldm/res_up_q"] = m.res_up(x, emb)
        d["ldm/tr_q7"] = m.tr(xs, ctx7)
        d["ldm/tr_q1"] = m.tr(xs, ctx1)
        d["ldm/ab_q"] = m.ab(x)
    for k, v in qparams_of(qh).items():
        d["ldm/" + k] = v
    save("g7_blocks", d)


class _ToyNet(nn.Module):
    """2-block toy model for G8/G12: conv_in -> ResnetBlock -> AttnBlock -> conv_out."""

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
        h = self.conv_in(x)
        h = self.rb(h, temb)
        h = self.at(h)
        return self.conv_out(h)


def g8_g12_recon():
    """G12: save_inp_oup_data nesting/values (data_utils.py:7-75,107-171).
    G8: layer_reconstruction / block_reconstruction trajectories with prob=1, input_prob=1
    (no device RNG; python `random` fixes idx) (layer_recon.py:13-129, block_recon.py:13-232)."""
    seed_everything(808)
    g = torch.Generator().manual_seed(808)
    net = _ToyNet().eval()
    d = {"sd/" + k: v for k, v in net.state_dict().items()}
    aq = dict(AQ8)
    aq["prob"] = 1.0
    qnn = QuantModel(net, WQ4, aq, sm_abit=8)
    qnn.eval()
    N = 64
    x = torch.randn(N, 3, 8, 8, generator=g)
    t = torch.randint(0, 1000, (N,), generator=g).float()
    d["x"], d["t"] = x, t
    cali = (x, t)
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali, batch_size=32)
    for k, v in qparams_of(qnn).items():
        d["init/" + k] = v
    # G12
    qnn.set_quant_state(True, True)
    res, ci, co = save_inp_oup_data(qnn, qnn.model.rb, cali, True, True, batch_size=32, input_prob=True, keep_gpu=False)
    d["g12/rb/resblock"] = np.int64(res)
    d["g12/rb/inp_q"], d["g12/rb/temb_q"] = ci[0][0], ci[0][1]
    d["g12/rb/inp_fp"], d["g12/rb/temb_fp"] = ci[1][0], ci[1][1]
    d["g12/rb/out_fp"] = co
    res, ci, co = save_inp_oup_data(qnn, qnn.model.conv_in, cali, True, True, batch_size=32, input_prob=True, keep_gpu=False)
    d["g12/conv_in/resblock"] = np.int64(res)
    d["g12/conv_in/inp_q"], d["g12/conv_in/inp_fp"], d["g12/conv_in/out_fp"] = ci[0], ci[1], co

    # G8: walk the units like recon_block_Qmodel, recording per-iteration state through a
    # patched Adam.step (alpha / delta after every iteration)
    kwargs = dict(cali_data=cali, iters=12, act_quant=True, asym=True, opt_mode="mse", lr_a=1e-3, lr_w=5e-2,
                  p=2.0, weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=1.0,
                  add_loss=0.8, recon_w=True, recon_a=True, keep_gpu=True)
    traj = {}
    cur = {"name": None}
    orig_step = torch.optim.Adam.step

    def step(self, *a, **k):
        r = orig_step(self, *a, **k)
        ps = [p for gr in self.param_groups for p in gr["params"]]
        key = "%s/%s" % (cur["name"], "a" if ps[0].numel() == 1 else "w")
        traj.setdefault(key, []).append(torch.cat([p.detach().flatten() for p in ps]).clone())
        return r

    torch.optim.Adam.step = step
    idx_log = {}
    orig_sample = random.sample

    def sample(pop, k):
        r = orig_sample(pop, k)
        idx_log.setdefault(cur["name"], []).append(list(r))
        return r

    random.sample = sample
    try:
        random.seed(8080)
        cur["name"] = "conv_in"
        layer_reconstruction(qnn, qnn.model.conv_in, **kwargs)
        cur["name"] = "temb_lin"
        layer_reconstruction(qnn, qnn.model.temb_lin, **kwargs)
        cur["name"] = "rb"
        block_reconstruction(qnn, qnn.model.rb, **kwargs)
        cur["name"] = "at"
        block_reconstruction(qnn, qnn.model.at, **kwargs)
        cur["name"] = "conv_out"
        layer_reconstruction(qnn, qnn.model.conv_out, **kwargs)
    finally:
        torch.optim.Adam.step = orig_step
        random.sample = orig_sample
    for k, v in traj.items():
        d["g8/traj/" + k] = torch.stack(v)
    for k, v in idx_log.items():
        d["g8/idx/" + k] = np.array(v)
    for k, v in qparams_of(qnn).items():
        d["g8/final/" + k] = v
    for name, m in qnn.named_modules():
        if isinstance(m, AdaRoundQuantizer):
            d["g8/final/alpha/" + name] = m.alpha
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        d["g8/final/out_q"] = qnn(x[:8], t[:8])
    d["g8/block_count"] = np.int64(qnn.block_count)
    save("g8_recon", d)


def _recon_variant(fname, prob, input_prob, N=32, iters=12):
    """G8b / G8c: the G8 walk (layer_recon.py:13-129, block_recon.py:13-232 on the 2-block toy model) with everything a test
    needs to REMOVE every cause of divergence but the loop itself: the reference's initial scales (init/qp/*), the cached
    (inp_q, inp_fp, out_fp) tensors of every unit as the reference's save_inp_oup_data returned them during ITS walk
    (data_utils.py:7-75), the drawn minibatch indices, and -- for prob / input_prob < 1, the shipped 0.5 / 0.5 of
    sample_diffusion_ldm_imagenet.py:144,185 -- the uniforms of every torch.rand_like call (block_recon.py:141-145 input
    mix; quant_layer.py:271-275 in both quantised forwards of an iteration, block_recon.py:154,167), supplied by
    tests/golden/_uniforms.py through a patched torch.rand_like and logged in call order."""
    import _uniforms
    import qdiff.block_recon as rb_mod
    import qdiff.layer_recon as rl_mod
    seed_everything(808)
    g = torch.Generator().manual_seed(808)
    net = _ToyNet().eval()
    d = {"sd/" + k: v for k, v in net.state_dict().items()}
    aq = dict(AQ8)
    aq["prob"] = prob
    qnn = QuantModel(net, WQ4, aq, sm_abit=8)
    qnn.eval()
    x = torch.randn(N, 3, 8, 8, generator=g)
    t = torch.randint(0, 1000, (N,), generator=g).float()
    d["x"], d["t"] = x, t
    d["prob"], d["input_prob"], d["iters"] = np.float64(prob), np.float64(input_prob), np.int64(iters)
    cali = (x, t)
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali, batch_size=32)
    for k, v in qparams_of(qnn).items():
        d["init/" + k] = v
    kwargs = dict(cali_data=cali, iters=iters, act_quant=True, asym=True, opt_mode="mse", lr_a=1e-3, lr_w=5e-2,
                  p=2.0, weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=input_prob,
                  add_loss=0.8, recon_w=True, recon_a=True, keep_gpu=True)
    traj, cur, idx_log = {}, {"name": None, "phase": "iter"}, {}
    rep = _uniforms.Replay()
    checks = []
    orig_step, orig_sample, orig_rand_like = torch.optim.Adam.step, random.sample, torch.rand_like

    def step(self, *a, **k):
        r = orig_step(self, *a, **k)
        ps = [p for gr in self.param_groups for p in gr["params"]]
        key = "%s/%s" % (cur["name"], "a" if ps[0].numel() == 1 else "w")
        traj.setdefault(key, []).append(torch.cat([p.detach().flatten() for p in ps]).clone())
        return r

    def sample(pop, k):
        r = orig_sample(pop, k)
        idx_log.setdefault(cur["name"], []).append(list(r))
        return r

    def rand_like(xx, **k):
        owner = sys._getframe(1).f_locals.get("self")
        name = None
        if owner is not None:
            for n, m in qnn.named_modules():
                if m is owner:
                    name = n
        if name is None:
            assert owner is None, type(owner)
            name = "input_mix:" + cur["name"]
        u = _uniforms.uniform(name, cur["phase"], rep.counts.get((name, cur["phase"]), 0), xx.shape)
        rep.draw(name, cur["phase"], xx.shape)
        checks.append([float(u.reshape(-1)[0]), float(u.reshape(-1)[-1]), float(u.astype(np.float64).sum())])
        return torch.from_numpy(u)

    def wrap_save(mod):
        orig = mod.save_inp_oup_data

        def run(*a, **k):
            cur["phase"] = "cache"
            try:
                res, ci, co = orig(*a, **k)
            finally:
                cur["phase"] = "iter"
            key = "cache/%s/" % cur["name"]
            d[key + "resblock"] = np.int64(res)
            if res:
                d[key + "inp_q"], d[key + "temb_q"] = ci[0][0], ci[0][1]
                d[key + "inp_fp"], d[key + "temb_fp"] = ci[1][0], ci[1][1]
            else:
                d[key + "inp_q"], d[key + "inp_fp"] = ci[0], ci[1]
            d[key + "out_fp"] = co
            return res, ci, co
        mod.save_inp_oup_data = run
        return orig

    torch.optim.Adam.step, random.sample, torch.rand_like = step, sample, rand_like
    osb, osl = wrap_save(rb_mod), wrap_save(rl_mod)
    units = (("conv_in", rl_mod.layer_reconstruction), ("temb_lin", rl_mod.layer_reconstruction),
             ("rb", rb_mod.block_reconstruction), ("at", rb_mod.block_reconstruction),
             ("conv_out", rl_mod.layer_reconstruction))
    try:
        random.seed(8080)
        for name, fn in units:
            cur["name"] = name
            fn(qnn, getattr(qnn.model, name), **kwargs)
    finally:
        torch.optim.Adam.step, random.sample, torch.rand_like = orig_step, orig_sample, orig_rand_like
        rb_mod.save_inp_oup_data, rl_mod.save_inp_oup_data = osb, osl
    for k, v in traj.items():
        d["traj/" + k] = torch.stack(v)
    for k, v in idx_log.items():
        d["idx/" + k] = np.array(v)
    for k, v in qparams_of(qnn).items():
        d["final/" + k] = v
    for name, m in qnn.named_modules():
        if isinstance(m, AdaRoundQuantizer):
            d["final/alpha/" + name] = m.alpha
            # index space: the hard-rounded integer weight codes the reference ends with (adaptive_rounding.py:49-60)
            w = dict(qnn.named_modules())[name.rsplit(".", 1)[0]].org_weight
            if name.endswith("_0"):
                raise AssertionError("toy model has no split layer")
            with torch.no_grad():
                xi = torch.floor(w / m.delta) + (m.alpha >= 0).float()
                d["final/codes/" + name] = torch.clamp(xi + m.zero_point, 0, m.n_levels - 1).to(torch.int16)
    d["rand/log"] = np.array(["%s|%s|%d|%s" % (o, p, c, "x".join(map(str, s))) for o, p, c, s in rep.log])
    d["rand/check"] = np.array(checks, dtype=np.float64).reshape(-1, 3)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        d["final/out_q"] = qnn(x[:8], t[:8])
    d["block_count"] = np.int64(qnn.block_count)
    save(fname, d)


def g8b_recon_masks():
    _recon_variant("g8b_recon_masks", 0.5, 0.5)


def g8c_recon_caches():
    _recon_variant("g8c_recon_caches", 1.0, 1.0)


class _FakeLDUncond(nn.Module):
    """Unconditional stand-in of LatentDiffusion (ddpm.py:895-997,1426-1445): schedules + apply_model, nothing else."""

    def __init__(self, net, linear_start=0.0015, linear_end=0.0195):
        super().__init__()

        class Wrap(nn.Module):
            def __init__(self, n):
                super().__init__()
                self.diffusion_model = n

        self.model = Wrap(net)
        self.num_timesteps = 1000
        b = make_beta_schedule("linear", 1000, linear_start=linear_start, linear_end=linear_end)
        ac = np.cumprod(1.0 - b, axis=0)
        self.betas = torch.tensor(b, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
        self.device = torch.device("cpu")

    def apply_model(self, x_, t_, c_):
        return self.model.diffusion_model(x_, t_, context=c_)


def g18_church_driver():
    """G18: config 3's named path end to end on the Church-shaped fixture network (weights of g13_ldm_church), in the order
    of sample_diffusion_ldm_church.py:256-311: TDAC_church_calib_data_generator (calibration.py:263-370, start noise and
    permutation captured) -> set_weight_quantize_params_LDM / set_act_quantize_params_LDM through
    DDIMSampler.sample(quant_unet=True, cali_data=[x, t, index]) (set_quantize_params_LDM.py:11-103, ddim.py:100-105,221-225)
    -> Change_LDM_model_attnblock -> the unconditional recon_block_Qmodel walk (recon_block_Qmodel.py:18-94) with the shipped
    kwargs (input_prob 0.5, quantizer prob 0.5, lr_w 5e-2, lr_a 1e-4, add_loss 1.0: for_church.sh) at 6 iterations per unit,
    uniforms through tests/golden/_uniforms.py, per-unit trajectories, and the cached tensors of a few representative units."""
    import _uniforms
    import scripts.calibration as refcal
    import qdiff.block_recon as rb_mod
    import qdiff.layer_recon as rl_mod
    from qdiff.set_quantize_params_LDM import set_weight_quantize_params_LDM, set_act_quantize_params_LDM
    rbq = sys.modules['qdiff.recon_block_Qmodel']
    base = np.load(os.path.join(HERE, "g13_ldm_church.npz"))
    kw = {k[4:]: (base[k].tolist() if base[k].ndim else base[k].item()) for k in base.files if k.startswith("cfg/")}
    seed_everything(1818)
    model = UNetModel(**kw).eval()
    model.load_state_dict({k[3:]: torch.as_tensor(base[k]) for k in base.files if k.startswith("sd/")})
    aq = dict(AQ8)                                  # leaf_param True, prob 0.5 (sample_diffusion_ldm_church.py:257)
    qnn = QuantModel(model, WQ4, aq, sm_abit=8)
    qnn.eval()
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(False, False)
    ld = _FakeLDUncond(qnn)
    N, nb, S = 32, 8, 20            # 20 steps: neighbouring early steps fall inside the density radius 0.3, later ones outside
    args = SimpleNamespace(custom_steps=S, eta=0.0, lamda=1.0)
    d = {"N": N, "nb": nb, "S": S, "lamda": 1.0}
    perms, draws = [], []
    orig_perm, orig_randn = torch.randperm, torch.randn
    torch.randperm = lambda n, **k: (perms.append(orig_perm(n, **k)) or perms[-1])

    def rec_randn(*a, **k):
        r = orig_randn(*a, **k)
        draws.append(r.clone())
        return r

    torch.randn = rec_randn
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        torch.manual_seed(1818)
        cali_data = refcal.TDAC_church_calib_data_generator(ld, args, N, nb, "cpu", S)
    finally:
        torch.randperm, torch.randn = orig_perm, orig_randn
        os.chdir(cwd)
    assert len(draws) == (N // nb) * (S + 1), len(draws)
    d["tdac/x_T"] = torch.stack([draws[i * (S + 1)] for i in range(N // nb)])
    d["tdac/perm"] = perms[-1]
    d["tdac/calib_data"], d["tdac/t"], d["tdac/index"] = cali_data
    qnn.model.split_shortcut = True
    set_weight_quantize_params_LDM(ld, cali_data, args)
    set_act_quantize_params_LDM(ld, cali_data, args, batch_size=16)        # two EMA batches
    for k, v in qparams_of(qnn).items():
        d["init/" + k] = v
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        d["init/out_q"] = qnn(cali_data[0][:8], cali_data[1][:8])
    Change_LDM_model_attnblock(qnn, aq)
    iters = 6
    kwargs = dict(cali_data=cali_data[:-1], iters=iters, act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-2,
                  p=2.0, weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=0.5, add_loss=1.0,
                  recon_w=True, recon_a=True, keep_gpu=False)
    d["iters"] = np.int64(iters)
    keep_caches = ("model.time_embed.0", "model.time_embed.2", "model.input_blocks.0.0", "model.input_blocks.1.0", "model.input_blocks.1.1", "model.input_blocks.2.0",
                   "model.output_blocks.0.0", "model.output_blocks.1.2", "model.out.2")
    traj, cur, idx_log, order = {}, {"name": None, "phase": "iter"}, {}, []
    rep = _uniforms.Replay()
    checks = []
    orig_step, orig_sample, orig_rand_like = torch.optim.Adam.step, random.sample, torch.rand_like

    def step(self, *a, **k):
        r = orig_step(self, *a, **k)
        ps = [p for gr in self.param_groups for p in gr["params"]]
        key = "%s/%s" % (cur["name"], "a" if ps[0].numel() == 1 else "w")
        traj.setdefault(key, []).append(torch.cat([p.detach().flatten() for p in ps]).clone())
        return r

    def sample(pop, k):
        r = orig_sample(pop, k)
        idx_log.setdefault(cur["name"], []).append(list(r))
        return r

    def rand_like(xx, **k):
        owner = sys._getframe(1).f_locals.get("self")
        name = None
        if owner is not None:
            for n, m in qnn.named_modules():
                if m is owner:
                    name = n
        if name is None:
            assert owner is None, type(owner)
            name = "input_mix:" + cur["name"]
        c = rep.counts.get((name, cur["phase"]), 0)
        u = rep.draw(name, cur["phase"], xx.shape)
        checks.append([float(u.reshape(-1)[0]), float(u.reshape(-1)[-1]), float(u.astype(np.float64).sum())])
        return torch.from_numpy(u)

    def wrap_save(mod):
        orig = mod.save_inp_oup_data

        def run(*a, **k):
            cur["phase"] = "cache"
            try:
                res, ci, co = orig(*a, **k)
            finally:
                cur["phase"] = "iter"
            if cur["name"] in keep_caches:
                key = "cache/%s/" % cur["name"]
                d[key + "resblock"] = np.int64(res)
                if res:
                    d[key + "inp_q"], d[key + "temb_q"] = ci[0][0], ci[0][1]
                    d[key + "inp_fp"], d[key + "temb_fp"] = ci[1][0], ci[1][1]
                else:
                    d[key + "inp_q"], d[key + "inp_fp"] = ci[0], ci[1]
                d[key + "out_fp"] = co
            return res, ci, co
        mod.save_inp_oup_data = run
        return orig

    ob, ol = rbq.block_reconstruction, rbq.layer_reconstruction

    def wrap_unit(kind, fn):
        def run(mdl, unit, **k2):
            names = {m: n for n, m in qnn.named_modules()}
            cur["name"] = names[unit]
            order.append("%s:%s:%s" % (kind, names[unit], type(unit).__name__))
            return fn(mdl, unit, **k2)
        return run

    torch.optim.Adam.step, random.sample, torch.rand_like = step, sample, rand_like
    osb, osl = wrap_save(rb_mod), wrap_save(rl_mod)
    rbq.block_reconstruction, rbq.layer_reconstruction = wrap_unit("block", ob), wrap_unit("layer", ol)
    try:
        random.seed(1818)
        qnn.set_quant_state(True, True)
        recon_block_Qmodel(args, qnn, cali_data, kwargs).recon()
    finally:
        torch.optim.Adam.step, random.sample, torch.rand_like = orig_step, orig_sample, orig_rand_like
        rb_mod.save_inp_oup_data, rl_mod.save_inp_oup_data = osb, osl
        rbq.block_reconstruction, rbq.layer_reconstruction = ob, ol
    d["order"] = np.array(order)
    for k, v in traj.items():
        # 834 k alphas: the first Adam step of every unit (the sign pattern of the first gradient) + the final state below;
        # the few step sizes at every iteration
        d["traj/" + k] = torch.stack(v) if k.endswith("/a") else v[0]
    for k, v in idx_log.items():
        d["idx/" + k] = np.array(v)
    for k, v in qparams_of(qnn).items():
        d["final/" + k] = v
    for name, m in qnn.named_modules():
        if isinstance(m, AdaRoundQuantizer):
            d["final/alpha/" + name] = m.alpha
    d["rand/log"] = np.array(["%s|%s|%d|%s" % (o, p, c, "x".join(map(str, s))) for o, p, c, s in rep.log])
    d["rand/check"] = np.array(checks, dtype=np.float64).reshape(-1, 3)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        d["final/out_q"] = qnn(cali_data[0][:8], cali_data[1][:8])
    d["block_count"] = np.int64(qnn.block_count)
    save("g18_church_driver", d)


def g19_tdac_others():
    """G19: the other TDAC generators end to end, values (calibration.py:12-155 cifar, :156-262 bedroom with the `> 0` fix-up,
    :502-638 coco through the PLMSSampler with classifier-free guidance), each on the matching fixture network with the
    start noise of every trajectory batch and the final permutation captured.  (Church: G18; ImageNet: G17.)"""
    import scripts.calibration as refcal
    from _weights import formula_state_dict
    d = {}

    def capture(fn):
        perms, draws = [], []
        orig_perm, orig_randn = torch.randperm, torch.randn
        torch.randperm = lambda n, **k: (perms.append(orig_perm(n, **k)) or perms[-1])

        def rec_randn(*a, **k):
            r = orig_randn(*a, **k)
            draws.append(r.clone())
            return r

        torch.randn = rec_randn
        cwd = os.getcwd()
        os.chdir("/tmp")
        try:
            out = fn()
        finally:
            torch.randperm, torch.randn = orig_perm, orig_randn
            os.chdir(cwd)
        return out, perms, draws

    # ---- cifar (DDPM Model, cali_generalized_steps; ONE trajectory batch serves every calibration batch: :100-118)
    base = np.load(os.path.join(HERE, "g13_cifar_unet.npz"))
    seed_everything(1919)
    cfg = cifar_cfg()
    model = DDPMModel(cfg).eval()
    model.load_state_dict({k[3:]: torch.as_tensor(base[k]) for k in base.files if k.startswith("sd/")})
    qnn = QuantModel(model, WQ4, dict(AQ8), sm_abit=8)
    qnn.eval()
    qnn.set_quant_state(False, False)
    S = 20
    seq = [int(v) for v in (np.linspace(0, np.sqrt(1000 * 0.8), S) ** 2)]          # the 'quad' skip of sample_diffusion_ddim.py:125-133
    betas = torch.linspace(1e-4, 2e-2, 1000)
    diffusion = SimpleNamespace(seq=seq, betas=betas, args=SimpleNamespace(eta=0.0))
    N, nb = 32, 16
    torch.manual_seed(1919)
    out, perms, draws = capture(lambda: refcal.TDAC_cifar_calib_data_generator(qnn.model, cfg, 1.2, N, nb, "cpu", diffusion, True))
    d["cifar/N"], d["cifar/nb"], d["cifar/lamda"], d["cifar/seq"] = N, nb, 1.2, np.array(seq)
    d["cifar/x_T"], d["cifar/perm"] = draws[0], perms[-1]
    d["cifar/calib_data"], d["cifar/t"], d["cifar/cls"] = out

    # ---- bedroom (unconditional LDM, `> 0` fix-up)
    baseh = np.load(os.path.join(HERE, "g13_ldm_church.npz"))
    kw = {k[4:]: (baseh[k].tolist() if baseh[k].ndim else baseh[k].item()) for k in baseh.files if k.startswith("cfg/")}
    net = UNetModel(**kw).eval()
    net.load_state_dict({k[3:]: torch.as_tensor(baseh[k]) for k in baseh.files if k.startswith("sd/")})
    qnn = QuantModel(net, WQ4, dict(AQ8), sm_abit=8)
    qnn.eval()
    qnn.set_quant_state(False, False)
    ld = _FakeLDUncond(qnn)
    N, nb, S = 32, 8, 20
    args = SimpleNamespace(custom_steps=S, eta=0.0, lamda=1.0)
    torch.manual_seed(1920)
    out, perms, draws = capture(lambda: refcal.TDAC_bedroom_calib_data_generator(ld, args, N, nb, "cpu", S))
    assert len(draws) == (N // nb) * (S + 1), len(draws)
    d["bedroom/N"], d["bedroom/nb"], d["bedroom/S"], d["bedroom/lamda"] = N, nb, S, 1.0
    d["bedroom/x_T"] = torch.stack([draws[i * (S + 1)] for i in range(N // nb)])
    d["bedroom/perm"] = perms[-1]
    d["bedroom/calib_data"], d["bedroom/t"], d["bedroom/index"] = out

    # ---- coco (Stable-Diffusion-shaped UNet, PLMS + classifier-free guidance)
    seed_everything(1921)
    g = torch.Generator().manual_seed(1921)
    net = UNetModel(**SD_KW).eval()
    sd = formula_state_dict([(k, v.shape) for k, v in net.state_dict().items()], 1305)
    net.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    qnn = QuantModel(net, WQ4, dict(AQ8), sm_abit=8, act_quant_mode="qdiff")
    qnn.eval()
    qnn.set_quant_state(False, False)
    qnn.set_grad_ckpt(False)
    N, nb, S = 8, 2, 30            # 30 PLMS steps: some neighbouring steps inside the density radius 0.3, most outside
    table = {"": torch.randn(77, 24, generator=g)}
    prompts = ["p%d" % i for i in range(N)]
    for p_ in prompts:
        table[p_] = torch.randn(77, 24, generator=g)
    ld = _FakeLDUncond(qnn, linear_start=0.00085, linear_end=0.012)
    ld.get_learned_conditioning = lambda ps: torch.stack([table[p_] for p_ in ps])       # CLIP stand-in: a lookup
    args = SimpleNamespace(custom_steps=S, scale=7.5, ddim_eta=0.0, plms=True, C=4, H=64, W=64, f=8, list_prompts=prompts, lamda=5.0)
    torch.manual_seed(1921)
    out, perms, draws = capture(lambda: refcal.TDAC_coco_calib_data_generator(ld, args, N, nb, "cpu", S))
    d["coco/N"], d["coco/nb"], d["coco/S"], d["coco/lamda"], d["coco/scale"] = N, nb, S, 5.0, 7.5
    d["coco/table"] = torch.stack([table[""]] + [table[p_] for p_ in prompts])
    d["coco/n_draws"] = len(draws)
    per = len(draws) // (N // nb)
    d["coco/x_T"] = torch.stack([draws[i * per] for i in range(N // nb)])
    d["coco/perm"] = perms[-1]
    for k, v in zip(("calib_data", "t", "index", "cond", "uncond", "t_next"), out):
        d["coco/" + k] = v
    save("g19_tdac_others", d)


def g9_tdac():
    """G9: TDAC scoring / allocation maths on synthetic feature maps (calibration.py:45-92;
    Church `>= 0` fix-up variant :332)."""
    g = torch.Generator().manual_seed(909)
    d = {}
    for T, N, lam, r in ((10, 64, 1.2, 3.0), (20, 256, 1.2, 3.0), (20, 100, 0.5, 2.0)):
        base = torch.randn(4, 32, 4, 4, generator=g)
        fm = [base * (0.3 + 0.2 * i) + torch.randn(4, 32, 4, 4, generator=g) * (0.2 + 0.25 * i) for i in range(T)]
        dense_num = torch.zeros(T, dtype=torch.int16)
        for i in range(T):
            for j in range(T):
                if i != j:
                    mse = torch.mean((fm[i] - fm[j]) ** 2)
                    if mse <= r:
                        dense_num[i] = dense_num[i] + 1
        dn = (dense_num - dense_num.min()) / (dense_num.max() - dense_num.min())
        cs = nn.CosineSimilarity(dim=1, eps=1e-6)
        cd = torch.zeros(T)
        for i in range(T):
            for j in range(T):
                if i != j:
                    cd[i] = cd[i] + torch.sum(1 - cs(fm[i], fm[j]))
        cdn = (cd - cd.min()) / (cd.max() - cd.min())
        w = dn + lam * cdn
        prob = w / torch.sum(w)
        key = "T%d_N%d" % (T, N)
        for variant in ("gt", "ge"):
            t_num = torch.tensor((prob * N).round(), dtype=int)
            t_error = N - torch.sum(t_num)
            _, t_num_sort = torch.sort(t_num, descending=True)
            if t_error >= 0:
                t_num[t_num_sort[:t_error]] += 1
            else:
                for i in reversed(range(len(t_num))):
                    if t_error == 0:
                        break
                    ok = (t_num[i] > 0) if variant == "gt" else (t_num[i] >= 0)
                    if ok:
                        t_num[i] -= 1
                        t_error = t_error + 1
            d[key + "/t_num_" + variant] = t_num
        d[key + "/fm"] = torch.stack(fm)
        d[key + "/lam"], d[key + "/r"], d[key + "/N"] = lam, r, N
        d[key + "/dense_num"], d[key + "/cos_dis"], d[key + "/w"] = dense_num, cd, w
    save("g9_tdac", d)


def g10_steps():
    """G10: stepping maths (denoising.py:4-59; util.py:21-74; ddim_control.py:198-254)."""
    g = torch.Generator().manual_seed(1010)
    d = {}
    betas = torch.linspace(1e-4, 0.02, 1000)
    tt = torch.tensor([0, 5, 500, 999])
    d["betas"], d["compute_alpha/t"], d["compute_alpha/a"] = betas, tt, compute_alpha(betas, tt)
    # one full generalized_steps run with a fixed linear "model"
    seq = [int(s) for s in (np.linspace(0, np.sqrt(1000 * 0.8), 10) ** 2)]
    x = torch.randn(4, 3, 8, 8, generator=g)
    Wm = torch.randn(3, 3, generator=g) * 0.3

    def model(xt, t):
        return torch.einsum("oc,bchw->bohw", Wm, xt) + (t.view(-1, 1, 1, 1) / 1000.0)

    xs, x0s = generalized_steps(x, seq, model, betas, eta=0.0)
    d["gs/seq"], d["gs/x"], d["gs/Wm"] = np.array(seq), x, Wm
    d["gs/xs"], d["gs/x0"] = torch.stack(xs), torch.stack(x0s)
    # eta=1 with injected noise
    noise = torch.randn(4, 3, 8, 8, generator=g)
    orig = torch.randn_like
    torch.randn_like = lambda t_, **k: noise
    try:
        xs1, _ = generalized_steps(x, seq[:3], model, betas, eta=1.0)
    finally:
        torch.randn_like = orig
    d["gs/noise"], d["gs/xs_eta1"] = noise, torch.stack(xs1)
    # LDM schedule helpers
    b = make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0195)
    ac = np.cumprod(1.0 - b, axis=0)
    d["ldm/betas"], d["ldm/alphas_cumprod"] = b, ac
    for S in (20, 50):
        ts = make_ddim_timesteps("uniform", S, 1000, verbose=False)
        sig, al, alp = make_ddim_sampling_parameters(ac, ts, 0.0, verbose=False)
        d["ldm/S%d/ts" % S], d["ldm/S%d/sigmas" % S] = ts, sig
        d["ldm/S%d/alphas" % S], d["ldm/S%d/alphas_prev" % S] = al, alp
    d["temb/t"] = torch.tensor([0., 1., 250., 999.])
    d["temb/ldm64"] = timestep_embedding(d["temb/t"], 64)
    from ddim.models.diffusion import get_timestep_embedding
    d["temb/ddpm64"] = get_timestep_embedding(d["temb/t"], 64)
    # p_sample_ddim through the real sampler with a stand-in LatentDiffusion (ddpm.py:895 contract)
    from ldm.models.diffusion.ddim_control import DDIMSampler_control

    class FakeLD:
        def __init__(self):
            self.num_timesteps = 1000
            self.betas = torch.tensor(b, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")

        def apply_model(self, x_, t_, c_):
            return torch.einsum("oc,bchw->bohw", Wm, x_) * 0.5 + c_.mean(dim=(1, 2)).view(-1, 1, 1, 1) \
                + t_.float().view(-1, 1, 1, 1) / 1000.0

    ld = FakeLD()
    s = DDIMSampler_control(ld)
    s.make_schedule(20, ddim_eta=0.0, verbose=False)
    c = torch.randn(4, 1, 16, generator=g)
    uc = torch.randn(4, 1, 16, generator=g)
    ts = torch.full((4,), int(s.ddim_timesteps[7]), dtype=torch.long)
    xp, px0 = s.p_sample_ddim(x, c, ts, index=7, unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
    d["ps/c"], d["ps/uc"], d["ps/x_prev"], d["ps/pred_x0"], d["ps/t"] = c, uc, xp, px0, ts
    idx = torch.tensor([0, 5, 11, 19])
    tq = torch.tensor(s.ddim_timesteps[idx.numpy()]).long()
    xp, px0 = s.p_sample_ddim(x, c, tq, index=idx, unconditional_guidance_scale=3.0, unconditional_conditioning=uc,
                              quant_unet=True)
    d["psq/index"], d["psq/t"], d["psq/x_prev"], d["psq/pred_x0"] = idx, tq, xp, px0
    save("g10_steps", d)


def g14_plms():
    """G14: the PLMS sampler (ldm/models/diffusion/plms.py:136-279): a full 8-step run through the reference's
    PLMSSampler (pseudo improved Euler first step, then Adams-Bashforth orders 2-4) with classifier-free guidance and
    a fixed linear stand-in for apply_model; every intermediate x and pred_x0 is kept."""
    from ldm.models.diffusion.plms import PLMSSampler
    g = torch.Generator().manual_seed(1414)
    b = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - b, axis=0)
    Wm = torch.randn(4, 4, generator=g) * 0.3

    class FakeLD:
        def __init__(self):
            self.num_timesteps = 1000
            self.betas = torch.tensor(b, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")

        def apply_model(self, x_, t_, c_):
            return torch.einsum("oc,bchw->bohw", Wm, x_) * 0.5 + c_.mean(dim=(1, 2)).view(-1, 1, 1, 1) \
                + t_.float().view(-1, 1, 1, 1) / 1000.0

    s = PLMSSampler(FakeLD())
    s.make_schedule(8, ddim_eta=0.0, verbose=False)
    x = torch.randn(3, 4, 8, 8, generator=g)
    c, uc = torch.randn(3, 2, 16, generator=g), torch.randn(3, 2, 16, generator=g)
    img, inter, _ = s.plms_sampling(c, (3, 4, 8, 8), x_T=x.clone(), unconditional_guidance_scale=7.5,
                                    unconditional_conditioning=uc)
    d = {"betas": b, "Wm": Wm, "x_T": x, "c": c, "uc": uc, "scale": np.float32(7.5), "ts": s.ddim_timesteps,
         "final": img, "x_inter": torch.stack(inter["x_inter"]), "pred_x0": torch.stack(inter["pred_x0"])}
    save("g14_plms", d)


def g15_decoder():
    """G15: the first-stage decoder (ldm/modules/diffusionmodules/model.py:465-572 Decoder with its ResnetBlock /
    AttnBlock / Upsample, plus the 1x1 post_quant_conv of VQModelInterface.decode, autoencoder.py:274-282) on a
    fixture-sized config: state dict, latent input, image output, and the output of the middle stage."""
    from ldm.modules.diffusionmodules.model import Decoder
    torch.manual_seed(1515)
    cfg = dict(ch=32, out_ch=3, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=[], dropout=0.0, in_channels=3,
               resolution=32, z_channels=3)
    dec = Decoder(**cfg).eval()
    pq = torch.nn.Conv2d(3, 3, 1)
    g = torch.Generator().manual_seed(15)
    with torch.no_grad():
        for prm in list(dec.parameters()) + list(pq.parameters()):
            prm.copy_(prm + 0.05 * torch.randn(prm.shape, generator=g))
        z = torch.randn(2, 3, 8, 8, generator=g)
        out = dec(pq(z))
        dec.give_pre_end = True
        pre = dec(pq(z))
    d = {"z": z, "out": out, "pre_end": pre}
    for k, v in cfg.items():
        d["cfg/" + k] = np.asarray(v)
    for k, v in dec.state_dict().items():
        d["sd/" + k] = v
    d["pq/weight"], d["pq/bias"] = pq.weight.detach(), pq.bias.detach()
    save("g15_decoder", d)


def g7b_blocks():
    """G7b: the Stable-Diffusion attention shapes at block level (quant_block.py:204-297,168-192; attention.py:218-287):
    8 heads, a 77-token context; an 8-head legacy AttentionBlock; a ResBlock whose skip convolution is split over the
    halves of a skip concatenation (quant_block.py:72-84)."""
    from ldm.modules.attention import SpatialTransformer
    seed_everything(7070)
    g = torch.Generator().manual_seed(7070)
    d = {}
    st = SpatialTransformer(64, 8, 8, depth=1, context_dim=48).eval()
    tr8 = BasicTransformerBlock(64, 8, 8, context_dim=48, checkpoint=False).eval()
    ab8 = AttentionBlock(64, num_heads=8).eval()
    res_split = ResBlock(64, 64, 0.0, out_channels=32).eval()
    holder = nn.Module()
    holder.in_channels = 3
    holder.st, holder.tr8, holder.ab8, holder.res_split = st, tr8, ab8, res_split
    reinit_zero_modules(holder, g)
    for k, v in holder.state_dict().items():
        d["sd/" + k] = v
    qh = QuantModel(holder, WQ4, AQ8, sm_abit=8)
    qh.set_grad_ckpt(False)
    x = torch.randn(2, 64, 4, 4, generator=g)
    xc = torch.randn(2, 64, 4, 4, generator=g) * torch.cat([torch.ones(32), 2.5 * torch.ones(32)]).view(1, 64, 1, 1)
    emb = torch.randn(2, 64, generator=g)
    xs = torch.randn(2, 16, 64, generator=g)
    ctx77 = torch.randn(2, 77, 48, generator=g)
    ctx1 = torch.randn(2, 1, 48, generator=g)
    d["x"], d["xc"], d["emb"], d["xs"], d["ctx77"], d["ctx1"] = x, xc, emb, xs, ctx77, ctx1
    m = qh.model
    assert isinstance(m.st.transformer_blocks[0], QuantBasicTransformerBlock) and isinstance(m.res_split, QuantResBlock)
    with torch.no_grad():
        d["st_fp"] = m.st(x, ctx77)
        d["tr8_fp77"], d["tr8_fp1"] = m.tr8(xs, ctx77), m.tr8(xs, ctx1)
        d["ab8_fp"] = m.ab8(x)
        d["res_split_fp"] = m.res_split(xc, emb, split=32)
        qh.set_quant_state(True, True)
        m.st(x, ctx77), m.tr8(xs, ctx77), m.ab8(x), m.res_split(xc, emb, split=32)      # initialises every quantizer
        for mm in qh.modules():
            if isinstance(mm, UniformAffineQuantizer):
                mm.set_inited(True)
        d["st_q"] = m.st(x, ctx77)
        d["tr8_q77"], d["tr8_q1"] = m.tr8(xs, ctx77), m.tr8(xs, ctx1)
        d["ab8_q"] = m.ab8(x)
        d["res_split_q"] = m.res_split(xc, emb, split=32)
    d.update(qparams_of(qh))
    save("g7b_blocks", d)


SD_KW = dict(image_size=8, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[1, 2],
             num_res_blocks=1, channel_mult=[1, 2], num_heads=8, use_spatial_transformer=True, transformer_depth=1,
             context_dim=24, legacy=False)


def g13_ldm_sd():
    """G13d: BASELINE config 5 in miniature -- a Stable-Diffusion-shaped UNetModel (v1-inference.yaml unet_config family:
    8 heads, 77-token context, transformer_depth 1, legacy False, NO split: sample_txt2img.py:183-184 sets an unused
    attribute) calibrated by the reference's own `set_{weight,act}_quantize_params_Stable` through its PLMSSampler
    (qdiff_control/set_quantize_params_Stable.py:12-145 with args.plms, plms.py:99-115,259-260), then FP / weight-quant /
    fake-quant forwards of the classifier-free-guidance batch.  Weights are formula weights (_weights.py), not stored."""
    from _weights import formula_state_dict
    from qdiff_control.set_quantize_params_Stable import (set_act_quantize_params_Stable,
                                                          set_weight_quantize_params_Stable)
    seed_everything(1305)
    g = torch.Generator().manual_seed(1305)
    model = UNetModel(**SD_KW).eval()
    sd = formula_state_dict([(k, v.shape) for k, v in model.state_dict().items()], 1305)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    d = {}
    for k, v in SD_KW.items():
        d["cfg/" + k] = np.asarray(v)
    d["weights_seed"] = np.int64(1305)
    N, S = 4, 8
    b = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - b, axis=0)
    ts = make_ddim_timesteps("uniform", S, 1000, verbose=False)
    index = torch.tensor([7, 5, 2, 0])
    x = torch.randn(N, 4, 8, 8, generator=g)
    t = torch.tensor(ts[index.numpy()]).long()
    t_next = torch.tensor(ts[np.maximum(index.numpy() - 1, 0)]).long()
    cond = torch.randn(N, 77, 24, generator=g)
    uncond = torch.randn(1, 77, 24, generator=g).expand(N, 77, 24).contiguous()
    cali = (x, t, index, cond, uncond, t_next)
    d["x"], d["t"], d["index"], d["cond"], d["uncond"], d["t_next"] = cali

    qnn = QuantModel(model, WQ4, AQ8, sm_abit=8, act_quant_mode="qdiff")
    qnn.eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_grad_ckpt(False)

    class Wrap(nn.Module):
        def __init__(self, net):
            super().__init__()
            self.diffusion_model = net

    class FakeLD(nn.Module):
        """stand-in for LatentDiffusion (ddpm.py:895-997 contract): schedules, apply_model, conditioning"""

        def __init__(self, net):
            super().__init__()
            self.model = Wrap(net)
            self.num_timesteps = 1000
            self.betas = torch.tensor(b, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")

        def apply_model(self, x_, t_, c_):
            return self.model.diffusion_model(x_, t_, context=c_)

        def get_learned_conditioning(self, prompts):
            return torch.zeros(len(prompts), 77, 24)          # CLIP stand-in: the PLMS calibration forward reads cali_data

    ld = FakeLD(qnn)
    args = SimpleNamespace(custom_steps=S, scale=7.5, ddim_eta=0.0, plms=True, C=4, H=64, W=64, f=8,
                           list_prompts=["a", "b", "c", "d"])
    for k in ("custom_steps", "scale", "ddim_eta", "C", "H", "W", "f"):
        d["args/" + k] = np.asarray(getattr(args, k))
    set_weight_quantize_params_Stable(ld, cali, args)
    set_act_quantize_params_Stable(ld, cali, args, batch_size=2)
    x8, t8, c8 = torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([uncond, cond])
    with torch.no_grad():
        qnn.set_quant_state(False, False)
        d["out_fp"] = qnn(x8, t8, c8)
        qnn.set_quant_state(True, False)
        d["out_wq"] = qnn(x8, t8, c8)
        qnn.set_quant_state(True, True)
        d["out_q"] = qnn(x8, t8, c8)
    d.update(qparams_of(qnn))
    n_split = sum(1 for m in qnn.modules() if isinstance(m, QuantModule) and m.split != 0)
    d["n_split_layers"] = np.int64(n_split)
    import qdiff_control.recon_block_Qmodel  # noqa: F401
    rbc = sys.modules['qdiff_control.recon_block_Qmodel']
    rec = []
    ob, ol = rbc.block_reconstruction, rbc.layer_reconstruction
    rbc.block_reconstruction = lambda m, blk, **k: rec.append(("block", blk))
    rbc.layer_reconstruction = lambda m, lay, **k: rec.append(("layer", lay))
    try:
        rbc.recon_block_Qmodel(None, qnn, cali, {}).recon()
    finally:
        rbc.block_reconstruction, rbc.layer_reconstruction = ob, ol
    names = {m: n for n, m in qnn.named_modules()}
    d["units"] = np.array(["%s:%s:%s" % (k, names[m], type(m).__name__) for k, m in rec])
    save("g13_ldm_sd", d)


def g16_layer_recon():
    """G16: the --layer_recon mode on the 2-block toy model: the reference's recon_layer_Qmodel walk
    (recon_layer_Qmodel.py:20-120: every QuantModule on its own, a QuantAttnBlock as q, k, v, its four attention step
    sizes alone via AttnBlock_layer_reconstruction, attn_layer_recon.py:13-133, then proj_out) with prob = input_prob = 1
    (no device RNG), 12 iterations per unit: unit order, alpha / delta after every Adam step, final state."""
    import qdiff.recon_layer_Qmodel  # noqa: F401
    rl = sys.modules['qdiff.recon_layer_Qmodel']
    seed_everything(1616)
    g = torch.Generator().manual_seed(1616)
    net = _ToyNet().eval()
    d = {"sd/" + k: v for k, v in net.state_dict().items()}
    aq = dict(AQ8)
    aq["prob"] = 1.0
    qnn = QuantModel(net, WQ4, aq, sm_abit=8)
    qnn.eval()
    N = 64
    x = torch.randn(N, 3, 8, 8, generator=g)
    t = torch.randint(0, 1000, (N,), generator=g).float()
    d["x"], d["t"] = x, t
    cali = (x, t)
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali, batch_size=32)
    for k, v in qparams_of(qnn).items():
        d["init/" + k] = v
    kwargs = dict(cali_data=cali, iters=12, act_quant=True, asym=True, opt_mode="mse", lr_a=1e-4, lr_w=5e-2,
                  p=2.0, weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=1.0,
                  add_loss=0.8, recon_w=True, recon_a=True, keep_gpu=True)
    names = {m: n for n, m in qnn.named_modules()}
    traj, order, cur = {}, [], {"name": None}
    orig_step = torch.optim.Adam.step

    def step(self, *a, **k):
        r = orig_step(self, *a, **k)
        ps = [p for gr in self.param_groups for p in gr["params"]]
        key = "%s/%s" % (cur["name"], "a" if ps[0].numel() == 1 else "w")
        traj.setdefault(key, []).append(torch.cat([p.detach().flatten() for p in ps]).clone())
        return r

    ol, oa = rl.layer_reconstruction, rl.AttnBlock_layer_reconstruction

    def wrap(kind, fn):
        def run(model, unit, **kw):
            cur["name"] = "%s:%s" % (kind, names[unit])
            order.append(cur["name"])
            return fn(model, unit, **kw)
        return run

    torch.optim.Adam.step = step
    rl.layer_reconstruction = wrap("layer", ol)
    rl.AttnBlock_layer_reconstruction = wrap("attn", oa)
    try:
        random.seed(1616)
        rl.recon_layer_Qmodel(None, qnn, cali, kwargs).recon()
    finally:
        torch.optim.Adam.step = orig_step
        rl.layer_reconstruction, rl.AttnBlock_layer_reconstruction = ol, oa
    d["order"] = np.array(order)
    for k, v in traj.items():
        d["traj/" + k] = torch.stack(v)
    for k, v in qparams_of(qnn).items():
        d["final/" + k] = v
    for name, m in qnn.named_modules():
        if isinstance(m, AdaRoundQuantizer):
            d["final/alpha/" + name] = m.alpha
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        d["final/out_q"] = qnn(x[:8], t[:8])
    d["block_count"] = np.int64(qnn.block_count)
    save("g16_layer_recon", d)


def g17_tdac_imagenet():
    """G17: the reference's TDAC_imagenet_calib_data_generator (scripts/calibration.py:371-500) end to end on the fixture
    LDM of g13_ldm_imagenet (its weights, FP state): DDIMSampler_control trajectories with classifier-free guidance, the
    mid-block features, density / variety scores, the allocation, the shuffled step assignment and the assembled
    calibration tuple.  Randomness is captured, not re-drawn: the start noise of every trajectory batch (x_inter[0]) and
    the permutation of torch.randperm are stored so that a test can inject them."""
    import scripts.calibration as refcal
    from qdiff.utils import AttentionMap  # noqa: F401
    base = np.load(os.path.join(HERE, "g13_ldm_imagenet.npz"))
    kw = {k[4:]: (base[k].tolist() if base[k].ndim else base[k].item()) for k in base.files if k.startswith("cfg/")}
    model = UNetModel(**kw).eval()
    model.load_state_dict({k[3:]: torch.as_tensor(base[k]) for k in base.files if k.startswith("sd/")})
    qnn = QuantModel(model, WQ4, AQ8, sm_abit=8, act_quant_mode="qdiff")
    qnn.eval()
    qnn.set_quant_state(False, False)
    b = make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0195)
    ac = np.cumprod(1.0 - b, axis=0)
    g = torch.Generator().manual_seed(1717)
    emb = torch.randn(1001, 16, generator=g) * 0.7

    class Wrap(nn.Module):
        def __init__(self, net):
            super().__init__()
            self.diffusion_model = net

    class FakeLD(nn.Module):
        def __init__(self, net):
            super().__init__()
            self.model = Wrap(net)
            self.num_timesteps = 1000
            self.cond_stage_key = "class_label"
            self.betas = torch.tensor(b, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)
            self.device = torch.device("cpu")

        def apply_model(self, x_, t_, c_):
            return self.model.diffusion_model(x_, t_, context=c_)

        def get_learned_conditioning(self, batch):
            return emb[batch[self.cond_stage_key]][:, None, :]          # ClassEmbedder (encoders/modules.py:21-33)

    ld = FakeLD(qnn)
    N, nb, S = 32, 8, 10
    labels = torch.randint(0, 1000, (N,), generator=g)
    args = SimpleNamespace(scale=3.0, data=labels, custom_steps=S, ddim_eta=0.0, lamda=1.2)
    perms = []
    orig_perm, orig_shape = torch.randperm, None
    torch.randperm = lambda n, **k: (perms.append(orig_perm(n, **k)) or perms[-1])
    draws = []
    orig_randn = torch.randn

    def rec_randn(*a, **k):
        r = orig_randn(*a, **k)
        draws.append(r.clone())
        return r

    torch.randn = rec_randn
    # the generator hard-codes the LDM-4 latent shape [3, 64, 64] (:380); the fixture UNet takes any H, W
    cwd = os.getcwd()
    os.chdir("/tmp")                                          # it also saves a plot into the working directory
    try:
        torch.manual_seed(1717)
        src = open(refcal.__file__).read()
        assert "shape = [3, 64, 64]" in src
        ns = dict(refcal.__dict__)
        exec(compile(src.replace("shape = [3, 64, 64]", "shape = [3, 8, 8]"), refcal.__file__, "exec"), ns)
        calib_data, t, index, cond, uncond = ns["TDAC_imagenet_calib_data_generator"](ld, args, N, nb, "cpu", S)
    finally:
        torch.randperm = orig_perm
        torch.randn = orig_randn
        os.chdir(cwd)
    d = {"emb": emb, "labels": labels, "N": N, "nb": nb, "S": S, "scale": 3.0, "lamda": 1.2, "perm": perms[-1],
         "calib_data": calib_data, "t": t, "index": index, "cond": cond, "uncond": uncond}
    # the start noise of each trajectory batch: the first torch.randn of every sampler.sample() call (each of the S steps
    # then draws one more through noise_like, multiplied by sigma = 0)
    assert len(draws) == (N // nb) * (S + 1), len(draws)
    d["x_T"] = torch.stack([draws[i * (S + 1)] for i in range(N // nb)])
    save("g17_tdac_imagenet", d)



class _G20Host(nn.Module):
    """holder of the two G20 units (tests/golden/_g20.py); never run as a network"""

    def __init__(self):
        super().__init__()
        import _g20
        self.in_channels = _g20.RES["channels"]
        self.res = ResBlock(_g20.RES["channels"], _g20.RES["emb_channels"], 0.0, out_channels=_g20.RES["out_channels"], dims=2,
                            use_checkpoint=False, use_scale_shift_norm=False)
        self.tf = BasicTransformerBlock(_g20.TF["dim"], _g20.TF["heads"], _g20.TF["d_head"], dropout=0.0,
                                        context_dim=_g20.TF["context_dim"], gated_ff=True, checkpoint=False)


def g20_f16x3_units(threads=None, fname="g20_f16x3_units", input_ulp=False, iters=None, fp64=False):
    """G20: qdiff_control.block_reconstruction (qdiff_control/block_recon.py:13-243) on ONE LDM-4-sized ResBlock (192 -> 384 at
    32 x 32) and ONE transformer block (d = 384, 1024 tokens), 32-row minibatches, the shipped ImageNet hyper-parameters and
    0.5 / 0.5 masks -- the size at which the product contracts on its three-product f16 kernels.  Weights, cached unit inputs
    and uniforms are formulas (tests/golden/_weights.py, _g20.py, _uniforms.uniform_hash); stored: the reference's initial
    scales, minibatch draws, the direction of every alpha's first Adam step, every alpha's final sign and whether it ends
    next to zero (bit-packed), strided trajectories, samples of the FP targets."""
    import time
    import _g20
    import _uniforms
    from _weights import formula_state_dict
    import qdiff_control.block_recon as cb_mod
    from qdiff_control.adaptive_rounding import AdaRoundQuantizer as AdaRoundControl
    ARQ = (AdaRoundQuantizer, AdaRoundControl)
    if threads:
        torch.set_num_threads(threads)          # the noise-floor run: same code, another partition of the fp32 sums
    seed_everything(_g20.SEED)
    host = _G20Host().eval()
    sd = formula_state_dict([(k, tuple(v.shape)) for k, v in host.state_dict().items()], _g20.SEED)
    host.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    if fp64:
        # the EXACT reference: the same fp32 weights, caches, scales, draws and masks, every operation of the reference's graph
        # evaluated in float64 (its one explicit fp32 cast, GroupNorm32.forward util.py:214-216, lifted for the run)
        from ldm.modules.diffusionmodules.util import GroupNorm32
        GroupNorm32.forward = nn.GroupNorm.forward
        torch.set_default_dtype(torch.float64)
        host = host.double()
    aq = dict(AQ8)
    aq["prob"] = _g20.PROB
    qnn = QuantModel(host, WQ4, aq, sm_abit=8)
    qnn.eval()
    qnn.set_grad_ckpt(False)
    d = {"weights_seed": np.int64(_g20.SEED), "iters": np.int64(_g20.ITERS)}
    traj, cur, idx_log = {}, {"name": None, "phase": "init"}, {}
    rep = _uniforms.ReplayHash()
    checks = []
    orig_step, orig_sample, orig_rand_like = torch.optim.Adam.step, random.sample, torch.rand_like

    grad0 = {}

    def step(self, *a, **k):
        ps = [p for gr in self.param_groups for p in gr["params"]]
        key = "%s/%s" % (cur["name"], "a" if ps[0].numel() == 1 else "w")
        if key not in grad0:                     # iteration 0: the gradients the first Adam step sees (block_recon.py:197-201)
            grad0[key] = torch.cat([p.grad.detach().flatten() for p in ps]).clone()
        r = orig_step(self, *a, **k)
        traj.setdefault(key, []).append(torch.cat([p.detach().flatten() for p in ps]).clone())
        return r

    def sample(pop, k):
        r = orig_sample(pop, k)
        idx_log.setdefault(cur["name"], []).append(list(r))
        return r

    def rand_like(xx, **k):
        owner = sys._getframe(1).f_locals.get("self")
        name = None
        if owner is not None:
            for n, m in qnn.named_modules():
                if m is owner:
                    name = n
        if name is None:
            assert owner is None, type(owner)
            name = "input_mix:" + cur["name"]
        u = rep.draw(name, cur["phase"], xx.shape)
        checks.append([float(u.reshape(-1)[0]), float(u.reshape(-1)[-1]), float(u.astype(np.float64).sum())])
        return torch.from_numpy(u)

    kwargs = dict(_g20.HYPER)
    kwargs.update(cali_data=None, iters=iters or _g20.ITERS)
    for name in ("res", "tf"):
        t0 = time.time()
        unit = getattr(qnn.model, name)
        cq, cf = _g20.caches(name)
        cq, cf = [torch.from_numpy(a) for a in cq], [torch.from_numpy(a) for a in cf]
        if fp64:
            cq, cf = [a.double() for a in cq], [a.double() for a in cf]
        if input_ulp:
            # the conditioning probe: every cached input moved by (at most) one unit in the last place, nothing else changed
            cq = [a * np.float32(1 + int(input_ulp) * 2.0 ** -23) for a in cq]
            cf = [a * np.float32(1 + int(input_ulp) * 2.0 ** -23) for a in cf]
        # initial scales the reference's way (set_quantize_params.py:48-69 / :9-46 on the unit): weight quantizers from one
        # forward in the (True, False) state, activation quantizers over two batches of 32 of the quantised-prefix inputs
        uaqs = [(n, m) for n, m in unit.named_modules() if isinstance(m, UniformAffineQuantizer)]
        unit.set_quant_state(True, False)
        for n, m in uaqs:
            if not m.leaf_param:
                m.set_inited(False)
        with torch.no_grad():
            unit(cq[0][:8], cq[1][:8])
        for n, m in uaqs:
            if not m.leaf_param:
                m.set_inited(True)
        unit.set_quant_state(True, True)
        for n, m in uaqs:
            if m.leaf_param:
                m.set_inited(False)
        with torch.no_grad():
            for i in range(_g20.ROWS // 32):
                unit(cq[0][i * 32:(i + 1) * 32], cq[1][i * 32:(i + 1) * 32])
        for n, m in uaqs:
            if m.leaf_param:
                m.set_inited(True)
        if input_ulp or fp64:
            # the probe moves the INPUTS only: the scales are the unperturbed run's (an MSE search over 100 candidates may pick
            # another candidate for a one-ulp change -- a 1 % step, not a rounding effect)
            base = np.load(os.path.join(HERE, "g20_f16x3_units.npz"))
            with torch.no_grad():
                for n, m in uaqs:
                    k = "init/qp/model.%s.%s" % (name, n)
                    if k + "/delta" not in base.files:           # a wrapper's own (unused, never initialised) quantizer
                        continue
                    for attr, key in (("delta", k + "/delta"), ("zero_point", k + "/zero_point")):
                        have = getattr(m, attr)
                        val = torch.as_tensor(base[key]).reshape(have.shape).to(have.dtype)
                        if isinstance(have, torch.nn.Parameter):
                            have.copy_(val)
                        else:
                            setattr(m, attr, val)
        print(name, "scales initialised %.0f s" % (time.time() - t0))
        for k, v in qparams_of(qnn).items():
            if k.startswith("qp/model.%s." % name):
                d["init/" + k] = v
        # FP targets of the cached rows (data_utils.py:133-139: the unit's output with quantisation off on the FP inputs)
        unit.set_quant_state(False, False)
        with torch.no_grad():
            out_fp = torch.cat([unit(cf[0][i:i + 32], cf[1][i:i + 32]) for i in range(0, _g20.ROWS, 32)])
        pos = _g20.sample_positions(out_fp.numel())
        d["out_fp/%s/sample" % name] = out_fp.reshape(-1)[torch.from_numpy(pos)]
        d["out_fp/%s/sum" % name] = np.float64(out_fp.double().sum())
        d["out_fp/%s/sumsq" % name] = np.float64((out_fp.double() ** 2).sum())

        def fake_save(model, block, cali, asym, act_quant, batch_size=32, input_prob=True, keep_gpu=True):
            return True, ([cq[0], cq[1]], [cf[0], cf[1]]), out_fp

        orig_save = cb_mod.save_inp_oup_data
        cb_mod.save_inp_oup_data = fake_save
        torch.optim.Adam.step, random.sample, torch.rand_like = step, sample, rand_like
        cur["name"], cur["phase"] = name, "iter"
        alpha0 = None
        try:
            random.seed(_g20.SEED + 1)
            t0 = time.time()
            cb_mod.block_reconstruction(qnn, unit, **kwargs)
            print(name, "reconstruction %.0f s" % (time.time() - t0))
        finally:
            torch.optim.Adam.step, random.sample, torch.rand_like = orig_step, orig_sample, orig_rand_like
            cb_mod.save_inp_oup_data = orig_save
        tw, ta = torch.stack(traj[name + "/w"]), torch.stack(traj[name + "/a"])
        # AdaRound's initial alpha in parameter order (adaptive_rounding.py:62-72), recomputed from the same scales
        a0 = []
        for n, m in unit.named_modules():
            if isinstance(m, ARQ):
                w = dict(unit.named_modules())[n.rsplit(".", 1)[0]].org_weight
                rest = (w / m.delta) - torch.floor(w / m.delta)
                a0.append((-torch.log((m.zeta - m.gamma) / (rest - m.gamma) - 1)).flatten())
        a0 = torch.cat(a0)
        assert a0.numel() == tw.shape[1]
        d["first/%s/up" % name] = _g20.pack((tw[0] > a0).numpy())           # direction of every alpha's first Adam step
        d["first/%s/moved" % name] = _g20.pack((tw[0] != a0).numpy())
        d["final/%s/sign" % name] = _g20.pack((tw[-1] >= 0).numpy())
        d["final/%s/near" % name] = _g20.pack((tw[-1].abs() < _g20.NEAR).numpy())
        d["final/%s/count" % name] = np.int64(tw.shape[1])
        d["traj/%s/w" % name] = tw[:, ::_g20.STRIDE].clone()
        d["traj/%s/a" % name] = ta
        # iteration-0 gradients: every STRIDE-th d loss / d alpha, the norm of all of them, every d loss / d delta
        gw, ga = grad0[name + "/w"], grad0[name + "/a"]
        d["grad0/%s/w" % name] = gw[::_g20.STRIDE].clone()
        d["grad0/%s/w_norm" % name] = np.float64(gw.double().norm())
        d["grad0/%s/w_nonzero" % name] = np.int64((gw != 0).sum())
        d["grad0/%s/a" % name] = ga.clone()
        d["idx/" + name] = np.array(idx_log[name])
        for k, v in qparams_of(qnn).items():
            if k.startswith("qp/model.%s." % name) and "act_quantizer" in k:
                d["final/" + k] = v
        # the parameter order of the trajectories (module paths of the AdaRound / trained activation quantizers)
        d["order/%s/w" % name] = np.array([n for n, m in unit.named_modules() if isinstance(m, ARQ)])
    d["rand/log"] = np.array(["%s|%s|%d|%s" % (o, p, c, "x".join(map(str, s))) for o, p, c, s in rep.log])
    d["rand/check"] = np.array(checks, dtype=np.float64).reshape(-1, 3)
    if threads:
        d = {k: v for k, v in d.items() if k.startswith(("final/res/", "final/tf/", "first/", "grad0/")) or (k.startswith("traj/") and k.endswith("/a"))}
        d["threads"] = np.int64(threads)
    if input_ulp:
        d = {k: v for k, v in d.items() if k.startswith("grad0/")}
        d["input_scale"] = np.float32(1 + int(input_ulp) * 2.0 ** -23)
    if fp64:
        torch.set_default_dtype(torch.float32)
        keep = ("grad0/",) if fp64 is True else ("grad0/", "final/res/", "final/tf/", "first/")
        d = {k: (v.double().numpy() if torch.is_tensor(v) else v) for k, v in d.items()
             if k.startswith(keep) and not k.startswith("final/qp")}
        d["dtype"] = np.array("float64")
    save(fname, d)


def g20_noise_floor():
    """The REFERENCE against ITSELF: the G20 run repeated with 3 CPU threads instead of 8 (torch splits its fp32 GEMM / convolution
    sums differently), same seeds, same formulas.  What differs between the two is the floor any other arithmetic -- the product's
    exact-fp32 or three-product f16 MFMA -- can be held to.  Stored next to G20 as the alternative run's bits."""
    g20_f16x3_units(threads=3, fname="g20_reference_3threads")


def g20_ulp_floor():
    """How well is the reference's OWN iteration-0 gradient defined?  The G20 run (scales initialised, then ONE iteration) with every
    cached unit input multiplied by 1 + 2^-23 -- a perturbation of one unit in the last place, the size of any fp32 rounding
    difference between two correct implementations.  The gradient is a function of pred - target through 8-bit fake-quantisers
    whose codes flip at .5 boundaries, so it is far less well conditioned than its inputs: the distance between the two REFERENCE
    gradients is the floor the product's gradient can be held to (tests/test_fullsize_gpu.py)."""
    g20_f16x3_units(fname="g20_reference_ulp", input_ulp=1, iters=1)


def g20_ulp32_floor():
    """the same probe with the inputs moved by 32 units in the last place (1 + 2^-18): the size of the difference between two correct
    fp32 evaluations of a 400..1500-term dot product in different summation orders.  The response grows like the square root of the
    perturbation (the number of flipped codes is proportional to it, their contributions add like noise)."""
    g20_f16x3_units(fname="g20_reference_ulp32", input_ulp=32, iters=1)


def g20_fp64_truth():
    """The EXACT value of the iteration-0 gradients: the reference's own graph (qdiff_control/block_recon.py:145-217,
    qdiff/quant_block.py:204-235) on the same fp32 weights, cached inputs, scales, minibatch draw and masks, every module and cached
    tensor in float64.  Neither the reference's fp32 CPU run nor the product's MFMA contraction is the truth: both are fp32-grade
    evaluations of a function whose 8-bit fake-quantisers flip codes at .5 boundaries; this run says how far EACH is from the value
    they approximate (tests/test_fullsize_gpu.py gates the product at a multiple of the reference's own distance)."""
    g20_f16x3_units(fname="g20_reference_fp64", iters=1, fp64=True)


def g20_fp64_full():
    """... and the whole 12-iteration run in float64 (same draws, masks, scales): every alpha's first-step direction and final hard
    rounding as exact arithmetic gives them.  The reference's fp32 run and the product are both compared with THIS in index space
    (tests/test_fullsize_gpu.py): the count by which the reference's own fp32 run misses the exact roundings is the yardstick."""
    g20_f16x3_units(fname="g20_reference_fp64_full", fp64="full")


if __name__ == "__main__":
    only = sys.argv[1:]
    jobs = dict(g1=g1_weight_init, g2=g2_act_init, g3=g3_uaq_forward, g4=g4_adaround, g5=g5_loss,
                g6=g6_quant_module, g7=g7_blocks, g8=g8_g12_recon, g9=g9_tdac, g10=g10_steps,
                g13c=g13_cifar_unet, g13i=lambda: g13_ldm_unet("imagenet"), g13h=lambda: g13_ldm_unet("church"), g14=g14_plms, g15=g15_decoder,
                g17=g17_tdac_imagenet, g7b=g7b_blocks, g13w8=lambda: g13_cifar_unet(8, "g13_cifar_w8", 1301), g13sd=g13_ldm_sd, g16=g16_layer_recon,
                g8b=g8b_recon_masks, g8c=g8c_recon_caches, g18=g18_church_driver, g19=g19_tdac_others, g1b=g1b_max_init, g20=g20_f16x3_units, g20n=g20_noise_floor, g20u=g20_ulp_floor, g20u32=g20_ulp32_floor, g20f64=g20_fp64_truth, g20f64full=g20_fp64_full)
    for k, fn in jobs.items():
        if not only or k in only:
            print("==", k)
            fn()
