"""Pins oracle/qdiff_oracle.py against vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import random

import numpy as np
import pytest
import torch

from oracle import qdiff_oracle as O

T = lambda a: torch.as_tensor(np.asarray(a))


def close(a, b, rtol=1e-6, atol=1e-7):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ---------------------------------------------------------------- G1
def test_g1_weight_init(golden):
    g = golden("g1_weight_init")
    keys = sorted({k.rsplit("/", 1)[0] for k in g.files if k.endswith("/delta")})
    assert len(keys) >= 12
    for key in keys:
        cname, b, s = key.split("/")
        w = T(g["w/" + cname])
        q = O.OQ(n_bits=int(b[1:]), symmetric=(s == "sym"), channel_wise=True)
        out = q(w)
        assert {"pos": 1, "neg": -1, "no": 0}[q.one_side_dist] == int(g[key + "/one_side"])
        np.testing.assert_array_equal(q.delta.numpy(), g[key + "/delta"])        # bit-exact
        np.testing.assert_array_equal(q.zero_point.numpy(), g[key + "/zero_point"])
        np.testing.assert_array_equal(out.numpy(), g[key + "/out"])


# ---------------------------------------------------------------- G2
def test_g2_act_init_ema(golden):
    g = golden("g2_act_init")
    runs = sorted({k.split("/step")[0] for k in g.files})
    for run in runs:
        cname, b, s = run.split("/")
        q = O.OQ(n_bits=int(b[1:]), symmetric=(s == "sym"), channel_wise=False, leaf_param=True, prob=0.5)
        for k in range(4):
            key = "%s/step%d" % (run, k)
            out = q(T(g[key + "/x"]))
            np.testing.assert_array_equal(q.delta.numpy(), g[key + "/delta"])
            np.testing.assert_array_equal(q.zero_point.numpy(), g[key + "/zero_point"])
            np.testing.assert_array_equal(q.running_min.numpy(), g[key + "/running_min"])
            np.testing.assert_array_equal(q.running_max.numpy(), g[key + "/running_max"])
            np.testing.assert_array_equal(out.numpy(), g[key + "/out"])


# ---------------------------------------------------------------- G3
def test_g3_fake_quant_fwd_bwd(golden):
    g = golden("g3_uaq_forward")
    x, gy, u = T(g["x"]), T(g["gy"]), T(g["mask_u"])
    cases = sorted({k.split("/")[0] for k in g.files if "/" in k})
    assert len(cases) == 4
    for c in cases:
        bits = int(c.split("_")[0][1:])
        delta = torch.tensor(float(c.split("_d")[1].split("_z")[0]))
        zp = torch.tensor(float(c.split("_z")[1]))
        out, codes = O.fake_quant_fwd(x, delta, zp, 2 ** bits)
        np.testing.assert_array_equal(codes.numpy(), g[c + "/codes"])
        np.testing.assert_array_equal(out.numpy(), g[c + "/out"])
        gx, gd = O.fake_quant_bwd(gy, x, delta, zp, 2 ** bits)
        np.testing.assert_array_equal(gx.numpy(), g[c + "/gx"])
        close(gd, g[c + "/gdelta"], rtol=2e-5)
        mask = u < 0.5
        out, _ = O.fake_quant_fwd(x, delta, zp, 2 ** bits, mask)
        np.testing.assert_array_equal(out.numpy(), g[c + "/train_out"])
        gx, gd = O.fake_quant_bwd(gy, x, delta, zp, 2 ** bits, mask)
        np.testing.assert_array_equal(gx.numpy(), g[c + "/train_gx"])
        close(gd, g[c + "/train_gdelta"], rtol=2e-5)


# ---------------------------------------------------------------- G4
def test_g4_adaround(golden):
    g = golden("g4_adaround")
    for c, bits in (("conv", 4), ("lin", 8)):
        w, delta, zp = T(g[c + "/w"]), T(g[c + "/delta"]), T(g[c + "/zero_point"])
        a0 = O.adaround_init_alpha(w, delta)
        np.testing.assert_array_equal(a0.numpy(), g[c + "/alpha0"])
        gy = T(g[c + "/gy"])
        close(O.adaround_fwd(w, a0, delta, zp, 2 ** bits, True), g[c + "/soft_out"], rtol=1e-6, atol=1e-7)
        close(O.adaround_bwd(gy, w, a0, delta, zp, 2 ** bits), g[c + "/galpha"], rtol=1e-5, atol=1e-9)
        a1 = T(g[c + "/alpha1"])
        close(O.adaround_fwd(w, a1, delta, zp, 2 ** bits, True), g[c + "/soft_out1"], rtol=1e-6, atol=1e-7)
        close(O.adaround_bwd(gy, w, a1, delta, zp, 2 ** bits), g[c + "/galpha1"], rtol=1e-5, atol=1e-9)
        np.testing.assert_array_equal(O.adaround_fwd(w, a1, delta, zp, 2 ** bits, False).numpy(), g[c + "/hard_out1"])


# ---------------------------------------------------------------- G5
def test_g5_loss_and_temp(golden):
    g = golden("g5_loss")
    for c in ("4d", "2d", "3d"):
        p, t = T(g[c + "/pred"]), T(g[c + "/tgt"])
        close(O.lp_loss(p, t), g[c + "/loss"], rtol=1e-6)
        close(O.lp_loss_grad(p, t), g[c + "/gpred"], rtol=1e-6, atol=1e-9)
        close(O.lp_loss(p, t, p=2.4, reduction="all"), g[c + "/loss_all"], rtol=1e-6)
    b = [O.linear_temp_decay(int(t), 100, 0.2, 20, 2) for t in g["temp/t"]]
    close(b, g["temp/b"], rtol=0, atol=0)


# ---------------------------------------------------------------- G6
def test_g6_quant_module(golden):
    g = golden("g6_quant_module")
    wq = dict(n_bits=4, symmetric=True, channel_wise=True, scale_method="mse")
    aq = dict(n_bits=8, symmetric=True, channel_wise=False, scale_method="mse", leaf_param=True, prob=0.5)
    geo = dict(conv3=("conv2d", 1, 1), conv3s2=("conv2d", 2, 0), conv1split=("conv2d", 1, 0),
               conv1d=("conv1d", 1, 0), linear=("linear", 1, 0), linear3d=("linear", 1, 0))
    for c, (kind, stride, pad) in geo.items():
        bias = T(g[c + "/bias"]) if c + "/bias" in g.files else None
        l = O.OLayer(c, kind, T(g[c + "/weight"]), bias, wq, aq, stride, pad)
        x = T(g[c + "/x"])
        split = int(g[c + "/split"])
        close(l(x, split) if split else l(x), g[c + "/out_fp"], rtol=1e-5, atol=1e-6)
        l.set_quant_state(True, False)
        close(l(x), g[c + "/out_w"], rtol=1e-5, atol=1e-6)
        for q in l.quantizers():
            if not q.leaf_param:
                q.inited = True
        l.set_quant_state(True, True)
        close(l(x), g[c + "/out_wa"], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(l.weight_quantizer.delta.numpy(), g[c + "/w_delta"])
        np.testing.assert_array_equal(l.act_quantizer.delta.numpy(), g[c + "/a_delta"])
        np.testing.assert_array_equal(l.act_quantizer.zero_point.numpy(), g[c + "/a_zp"])
        if split:
            np.testing.assert_array_equal(l.weight_quantizer_0.delta.numpy(), g[c + "/w_delta_0"])
            np.testing.assert_array_equal(l.act_quantizer_0.delta.numpy(), g[c + "/a_delta_0"])


# ---------------------------------------------------------------- G9
def test_g9_tdac(golden):
    g = golden("g9_tdac")
    for key in sorted({k.split("/")[0] for k in g.files}):
        fm = list(T(g[key + "/fm"]))
        for variant in ("gt", "ge"):
            dense, cd, w, t_num = O.tdac_allocate(fm, float(g[key + "/lam"]), int(g[key + "/N"]), float(g[key + "/r"]),
                                                  fixup_ge=(variant == "ge"))
            np.testing.assert_array_equal(dense.numpy(), g[key + "/dense_num"])
            close(cd, g[key + "/cos_dis"], rtol=1e-5)
            np.testing.assert_array_equal(t_num.numpy(), g[key + "/t_num_" + variant])
            assert int(t_num.sum()) == int(g[key + "/N"])


# ---------------------------------------------------------------- G10
def test_g10_stepping(golden):
    g = golden("g10_steps")
    betas = T(g["betas"])
    close(O.compute_alpha(betas, T(g["compute_alpha/t"])), g["compute_alpha/a"], rtol=1e-6)
    Wm = T(g["gs/Wm"])
    model = lambda xt, t: torch.einsum("oc,bchw->bohw", Wm, xt) + (t.view(-1, 1, 1, 1) / 1000.0)
    seq = [int(s) for s in g["gs/seq"]]
    xs, x0s = O.generalized_steps(T(g["gs/x"]), seq, model, betas, 0.0)
    close(torch.stack(xs), g["gs/xs"], rtol=1e-5, atol=1e-6)
    close(torch.stack(x0s), g["gs/x0"], rtol=1e-5, atol=1e-6)
    xs1, _ = O.generalized_steps(T(g["gs/x"]), seq[:3], model, betas, 1.0, noise=T(g["gs/noise"]))
    close(torch.stack(xs1), g["gs/xs_eta1"], rtol=1e-5, atol=1e-6)
    b = O.ldm_linear_betas(1000, 0.0015, 0.0195)
    close(b, g["ldm/betas"], rtol=0, atol=0)
    ac = np.cumprod(1.0 - b, axis=0)
    for S in (20, 50):
        ts = O.make_ddim_timesteps(S, 1000)
        np.testing.assert_array_equal(ts, g["ldm/S%d/ts" % S])
        sig, al, alp = O.make_ddim_sampling_parameters(ac, ts, 0.0)
        close(al, g["ldm/S%d/alphas" % S], rtol=0, atol=0)
        close(alp, g["ldm/S%d/alphas_prev" % S], rtol=0, atol=0)
    close(O.ldm_timestep_embedding(T(g["temb/t"]), 64), g["temb/ldm64"], rtol=1e-6, atol=1e-7)
    close(O.ddpm_timestep_embedding(T(g["temb/t"]), 64), g["temb/ddpm64"], rtol=1e-6, atol=1e-7)
    # CFG p_sample_ddim with the stand-in apply_model of the fixture
    x, c, uc = T(g["gs/x"]), T(g["ps/c"]), T(g["ps/uc"])
    app = lambda x_, t_, c_: torch.einsum("oc,bchw->bohw", Wm, x_) * 0.5 + c_.mean(dim=(1, 2)).view(-1, 1, 1, 1) \
        + t_.float().view(-1, 1, 1, 1) / 1000.0
    ts20 = O.make_ddim_timesteps(20, 1000)
    sig, al, alp = O.make_ddim_sampling_parameters(ac.astype(np.float32), ts20, 0.0)
    f = lambda v: torch.full((4, 1, 1, 1), float(v))
    t = T(g["ps/t"])
    xp, px0 = O.p_sample_ddim(x, app(x, t, c), app(x, t, uc), 3.0, f(al[7]), f(alp[7]), f(sig[7]),
                              f(np.sqrt(1 - al[7])))
    close(xp, g["ps/x_prev"], rtol=2e-5, atol=2e-6)
    close(px0, g["ps/pred_x0"], rtol=2e-5, atol=2e-6)
    idx = g["psq/index"]
    tq = T(g["psq/t"])
    v = lambda a: torch.tensor(np.asarray(a)[idx], dtype=torch.float32).view(-1, 1, 1, 1)
    xp, px0 = O.p_sample_ddim(x, app(x, tq, c), app(x, tq, uc), 3.0, v(al), v(alp), v(sig), v(np.sqrt(1 - al)))
    close(xp, g["psq/x_prev"], rtol=2e-5, atol=2e-6)
    close(px0, g["psq/pred_x0"], rtol=2e-5, atol=2e-6)


def test_g14_plms(golden):
    """the oracle's PLMS restatement against a full run of the reference's PLMSSampler (orders 1-4 + the pseudo
    improved Euler first step, CFG 7.5)"""
    g = golden("g14_plms")
    Wm = T(g["Wm"])
    app = lambda x_, t_, c_: torch.einsum("oc,bchw->bohw", Wm, x_) * 0.5 + c_.mean(dim=(1, 2)).view(-1, 1, 1, 1) \
        + t_.float().view(-1, 1, 1, 1) / 1000.0
    ac = np.cumprod(1.0 - g["betas"], axis=0)
    final, xs, x0s = O.plms_sample(app, T(g["x_T"]), T(g["c"]), T(g["uc"]), float(g["scale"]), ac, 8)
    close(torch.stack(xs), g["x_inter"][1:], rtol=2e-5, atol=2e-5)
    close(torch.stack(x0s), g["pred_x0"][1:], rtol=2e-5, atol=2e-5)
    close(final, g["final"], rtol=2e-5, atol=2e-5)
