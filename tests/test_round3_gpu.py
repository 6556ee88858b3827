"""-m gpu, round 3: the parity holes the round-2 review named.

* The reconstruction loop in INDEX SPACE: the reference's initial scales loaded (not re-derived), the reference's cached unit
  inputs / outputs injected, the reference's minibatch draws -- what is left is the loop itself (K1 / K2 / K7 / K8 / K10 and
  the three-product contraction) and the final hard rounding must agree, every disagreeing weight listed with its alpha.
* The same loop with the SHIPPED stochastic setting, prob = input_prob = 0.5 (sample_diffusion_ldm_imagenet.py:144,185):
  the uniforms the reference consumed (block_recon.py:141-145, quant_layer.py:271-275, both quantised forwards) replayed
  through `injected_uniform` / `recon.INJECT_MIX_UNIFORM`.
"""
import os
import random
import sys

import numpy as np
import pytest
import torch
from types import SimpleNamespace

from helpers import build_toynet, WQ4, AQ8

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import _uniforms  # noqa: E402

pytestmark = pytest.mark.gpu

UNITS = (("conv_in", "layer"), ("temb_lin", "layer"), ("rb", "block"), ("at", "block"), ("conv_out", "layer"))


def _cuda(a):
    return torch.as_tensor(np.asarray(a)).cuda()


@pytest.mark.parametrize("fixture", ["g8c_recon_caches", "g8b_recon_masks"])
def test_recon_loop_index_space_reference_scales_caches_masks(golden, fixture):
    from qdiff import QuantModel
    from qdiff.block_recon import block_reconstruction
    from qdiff.layer_recon import layer_reconstruction
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.quant_layer import UniformAffineQuantizer
    from edadm.state import load_quant_state
    import edadm.recon as recon
    g = golden(fixture)
    prob, input_prob, iters = float(g["prob"]), float(g["input_prob"]), int(g["iters"])
    aq = dict(AQ8)
    aq["prob"] = prob
    qnn = QuantModel(build_toynet(g), WQ4, aq, sm_abit=8).cuda().eval()
    x, t = _cuda(g["x"]), _cuda(g["t"])
    n = load_quant_state(qnn, {k: g[k] for k in g.files if k.startswith("init/qp/")}, prefix="init/qp/")
    assert n == len([k for k in g.files if k.startswith("init/qp/") and k.endswith("/delta")])
    rep = _uniforms.Replay()
    from qdiff.quant_block import QuantAttnBlock
    mods = dict(qnn.named_modules())
    for name, m in qnn.named_modules():
        if isinstance(m, UniformAffineQuantizer) and m.leaf_param:
            # the product keeps the attention probabilities as p[b, i, j]; the reference quantises w_.permute(0, 2, 1)
            # (quant_block.py:436-441): a per-tensor quantiser, same codes -- the mask element of (i, j) is the reference's (j, i)
            tr = name.endswith(".act_quantizer_w") and isinstance(mods[name.rsplit(".", 1)[0]], QuantAttnBlock)
            m.injected_uniform = (lambda nm, tr: lambda xx: torch.from_numpy(
                rep.draw_calls(nm, "iter", xx.shape, recon.STATE["batched"]).transpose(0, 2, 1).copy() if tr
                else rep.draw_calls(nm, "iter", xx.shape, recon.STATE["batched"])).to(xx.device))(name, tr)
    cur = {"name": None}
    recon.INJECT_MIX_UNIFORM = lambda xx: torch.from_numpy(rep.draw("input_mix:" + cur["name"], "iter", xx.shape)).to(xx.device)

    def golden_save_fn(model, unit, cali, asym, act_quant, batch_size=32, input_prob=True, keep_gpu=True):
        k = "cache/%s/" % cur["name"]
        if bool(g[k + "resblock"]):
            return True, ([_cuda(g[k + "inp_q"]), _cuda(g[k + "temb_q"])], [_cuda(g[k + "inp_fp"]), _cuda(g[k + "temb_fp"])]), \
                _cuda(g[k + "out_fp"])
        return False, (_cuda(g[k + "inp_q"]), _cuda(g[k + "inp_fp"])), _cuda(g[k + "out_fp"])

    kwargs = dict(cali_data=(x, t), iters=iters, act_quant=True, asym=True, opt_mode="mse", lr_a=1e-3, lr_w=5e-2, p=2.0,
                  weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=input_prob, add_loss=0.8,
                  recon_w=True, recon_a=True, keep_gpu=True)
    traj = {}
    orig = recon.FusedAdam.launch

    def step(self):
        orig(self)
        key = "%s/%s" % (cur["name"], "a" if self.params[0].numel() == 1 else "w")
        traj.setdefault(key, []).append(self.flat.detach().cpu().clone())

    recon.FusedAdam.launch = step
    idx_log, orig_sample = {}, random.sample

    def sample(pop, k):
        r = orig_sample(pop, k)
        idx_log.setdefault(cur["name"], []).append(list(r))
        return r

    random.sample = sample
    try:
        random.seed(8080)
        for name, kind in UNITS:
            cur["name"] = name
            recon.reconstruct(qnn, getattr(qnn.model, name), kwargs["cali_data"], is_block=(kind == "block"),
                              save_fn=golden_save_fn, **{k: v for k, v in kwargs.items() if k != "cali_data"})
            assert np.array_equal(np.asarray(idx_log[name]), g["idx/" + name])      # the reference's minibatch draws
    finally:
        random.sample = orig_sample
        recon.FusedAdam.launch = orig
        recon.INJECT_MIX_UNIFORM = None
    # every draw of the iterations consumed in the reference's per-quantizer order and shape
    ref_log = sorted(l for l in g["rand/log"] if "|iter|" in l)
    got_log = sorted("%s|%s|%d|%s" % (o, p, c, "x".join(map(str, s))) for o, p, c, s in rep.log)
    assert got_log == ref_log
    if os.environ.get("EDADM_TEST_DUMP"):
        np.savez(os.path.join(os.environ["EDADM_TEST_DUMP"], "traj_%s.npz" % fixture),
                 **{k.replace("/", "_"): torch.stack(v).numpy() for k, v in traj.items()})
    stats = []
    for name, _ in UNITS:
        ref_w, ref_a = g["traj/%s/w" % name], g["traj/%s/a" % name]
        got_w, got_a = torch.stack(traj[name + "/w"]).numpy(), torch.stack(traj[name + "/a"]).numpy()
        dw = np.abs(got_w - ref_w)
        da = np.abs(got_a - ref_a)
        print(fixture, name, "alpha vs REFERENCE: median %.2e frac>1e-2 %.5f max %.3g | delta max abs %.3g rel %.3g" % (
            np.median(dw), (dw > 1e-2).mean(), dw.max(), da.max(), (da / np.abs(ref_a)).max()))
        # identical inputs, scales, draws and masks: what differs is fp32 summation order (GPU three-product contraction vs
        # CPU), which Adam's normalisation amplifies only where a gradient is at rounding-noise level
        stats.append((name, np.median(dw), (dw > 1e-2).mean(), dw.max(), got_a, ref_a))
    # Blocks at prob = 1: every activation goes through a rounding, the GPU's fp32 sums differ from the CPU's in the last bit,
    # and about once in a few iterations ONE activation code lands on the other side of a rounding boundary (about 1e-5 per
    # element and forward).  In this toy ResnetBlock (GroupNorm with one channel per group: gradients are what survives the
    # projection, 1e-6) such a flip moves one output channel's alphas by a fraction of a step and Adam carries it on:
    # test_recon_iteration_gradients_teacher_forced below shows the per-iteration gradients agreeing to 4e-6 of their maximum
    # except in exactly such an iteration, where the deviation is confined to one output channel.
    for name, med, frac, mx, got_a, ref_a in stats:
        assert med < 5e-4, (name, med)
        assert frac < (2.5e-2 if name == "rb" else 5e-3) and mx < 2 * 5e-2, (name, frac, mx)
        np.testing.assert_allclose(got_a, ref_a, rtol=5e-3, atol=6e-4 if name == "at" else 1e-6)
    # index space: final hard rounding of every weight against the reference's, disagreements listed
    bad, total = [], 0
    mods = dict(qnn.named_modules())
    for name, m in qnn.named_modules():
        if isinstance(m, AdaRoundQuantizer):
            ref_alpha = g["final/alpha/" + name]
            got = m.alpha.detach().cpu().numpy()
            total += got.size
            for i in zip(*np.nonzero((got >= 0) != (ref_alpha >= 0))):
                bad.append((name, tuple(int(v) for v in i), float(ref_alpha[i]), float(got[i])))
            # the integer codes the frozen model would carry
            w = mods[name.rsplit(".", 1)[0]].org_weight
            with torch.no_grad():
                codes = torch.clamp(torch.floor(w / m.delta) + (m.alpha >= 0).float() + m.zero_point, 0, m.n_levels - 1)
            ndiff = int((codes.cpu().numpy().astype(np.int16) != g["final/codes/" + name]).sum())
            assert ndiff == sum(1 for b in bad if b[0] == name), (name, ndiff)
    print(fixture, "hard-rounding disagreements with the reference: %d of %d" % (len(bad), total), bad)
    # Agreement except weights whose alpha ENDS within a fifth of one Adam step (lr_w = 5e-2) of zero in BOTH runs -- the
    # CPU oracle against the reference leaves 0 (masks) / 4 (prob 1) of these 26 816, the GPU 2 / 15 (measured, round 3): a
    # handful of alphas that twelve +-lr steps park next to the rounding boundary, decided by the last bits of a gradient
    # gate: 2x the measurement (round 4: 15 at prob 1, 2 with the shipped masks)
    assert all(abs(r) < 1e-2 and abs(o) < 1e-2 for _, _, r, o in bad) and len(bad) <= (30 if fixture == "g8c_recon_caches" else 4), bad


@pytest.mark.parametrize("fixture,unit", [("g8c_recon_caches", "rb"), ("g8c_recon_caches", "at"), ("g8b_recon_masks", "rb"),
                                          ("g8b_recon_masks", "at"), ("g8b_recon_masks", "conv_in")])
def test_recon_iteration_gradients_teacher_forced(fixture, unit):
    """One iteration of the loop at a time, product (HIP) and oracle (CPU, pinned to the reference by
    tests/test_oracle_round3.py) both FORCED to the reference's alphas / step sizes after iteration k - 1, on the reference's
    caches, minibatch and masks: the gradients of all alphas and step sizes agree to 5e-5 of the largest one (measured 1e-6 to
    1.4e-5).  Iterations in which an activation code flips (see above; the prob = 1 ResnetBlock shows two in four) must still
    agree to 5 % of the largest gradient (measured 1.1 %, confined to one output channel)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import recon_grad_check
    res = recon_grad_check.main(fixture, unit, [0, 1, 2, 3])
    dirty = sorted({k for k, key, err, med in res if err > 5e-5})
    print(fixture, unit, [(k, key, "%.2e" % err) for k, key, err, med in res], "iterations with a flip:", dirty)
    assert len(dirty) <= 2 and all(err < 5e-2 for _, _, err, _ in res), res


def _church_ld(golden):
    from edadm.latent import LatentDiffusionLite
    from qdiff import QuantModel
    from helpers import build_ldm
    base = golden("g13_ldm_church")
    qnn = QuantModel(build_ldm(base), WQ4, AQ8, sm_abit=8).cuda().eval()      # leaf_param True, prob 0.5: the shipped dicts
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    qnn.set_quant_state(False, False)
    ld = LatentDiffusionLite(qnn, timesteps=1000, linear_start=0.0015, linear_end=0.0195, conditioning_key=None).cuda()
    return qnn, ld


def test_church_config3_tdac_and_ldm_scale_init_driver(golden):
    """Config 3's named path, first half (sample_diffusion_ldm_church.py:271-283): TDAC_church_calib_data_generator
    (calibration.py:263-370, the `>= 0` fix-up variant; the reference's start noise and permutation injected) VALUES, then
    set_weight_quantize_params_LDM / set_act_quantize_params_LDM (set_quantize_params_LDM.py:11-103) driven through
    DDIMSampler.sample(quant_unet=True, cali_data=[x, t, index]) on the reference's calibration tuple: weight step sizes
    and zero points bit-exact, activation step sizes within the flat-minimum grid tolerance, quantised output."""
    from scripts.calibration import TDAC_church_calib_data_generator
    from qdiff import set_weight_quantize_params_LDM, set_act_quantize_params_LDM
    from qdiff.quant_layer import UniformAffineQuantizer
    g = golden("g18_church_driver")
    qnn, ld = _church_ld(golden)
    N, nb, S = int(g["N"]), int(g["nb"]), int(g["S"])
    args = SimpleNamespace(custom_steps=S, eta=0.0, lamda=float(g["lamda"]))
    xT = iter(torch.as_tensor(g["tdac/x_T"]))
    orig_randn, orig_perm = torch.randn, torch.randperm
    torch.randn = lambda *a, **k: next(xT).cuda()
    torch.randperm = lambda n, **k: torch.as_tensor(g["tdac/perm"])
    try:
        calib, t, index = TDAC_church_calib_data_generator(ld, args, N, nb, torch.device("cuda"), S)
    finally:
        torch.randn, torch.randperm = orig_randn, orig_perm
    np.testing.assert_array_equal(t.cpu().numpy(), g["tdac/t"])
    np.testing.assert_array_equal(index.cpu().numpy(), g["tdac/index"])
    err = np.abs(calib.cpu().numpy() - g["tdac/calib_data"]).max() / np.abs(g["tdac/calib_data"]).max()
    print("TDAC church calibration latents vs the reference generator: max %.2e of range" % err)
    assert err <= 1e-4
    # the drivers on the REFERENCE's tuple (so that the scales are compared on identical inputs)
    cali = (_cuda(g["tdac/calib_data"]), _cuda(g["tdac/t"]), _cuda(g["tdac/index"]))
    qnn.model.split_shortcut = True
    set_weight_quantize_params_LDM(ld, cali, args)
    set_act_quantize_params_LDM(ld, cali, args, batch_size=16)
    n, worst = 0, 0.0
    for name, m in qnn.named_modules():
        if isinstance(m, UniformAffineQuantizer) and m.delta is not None:
            k = "init/qp/" + name
            assert k + "/delta" in g.files, k
            ref_d, ref_z = g[k + "/delta"].reshape(-1), g[k + "/zero_point"].reshape(-1)
            got_d, got_z = m.delta.detach().cpu().numpy().reshape(-1), m.zero_point.cpu().numpy().reshape(-1)
            assert m.n_bits == int(g[k + "/n_bits"]) and m.inited, k
            if m.leaf_param:
                worst = max(worst, float(np.abs(got_d / ref_d - 1).max()))
                np.testing.assert_allclose(got_d, ref_d, rtol=0.1)
                assert np.abs(got_z - ref_z).max() <= 1, k
            else:
                np.testing.assert_array_equal(got_d, ref_d)
                np.testing.assert_array_equal(got_z, ref_z)
            n += 1
    assert n == len([k for k in g.files if k.startswith("init/qp/") and k.endswith("/delta")])
    print("activation step sizes vs reference: worst %.3f" % worst)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        out = qnn(cali[0][:8], cali[1][:8]).cpu().numpy()
    err = np.abs(out - g["init/out_q"]) / np.abs(g["init/out_q"]).max()
    # the first difference in an operand is a +-1 code (census test of round 2); a random 4-bit network spreads it
    assert err.max() < 0.15 and err.mean() < 0.02, (err.max(), err.mean())


def test_church_config3_unconditional_walk_with_shipped_masks(golden):
    """Config 3's named path, second half (sample_diffusion_ldm_church.py:285-311): Change_LDM_model_attnblock + the
    unconditional recon_block_Qmodel walk with the shipped kwargs (input_prob 0.5, quantizer prob 0.5), started from the
    reference's scales, its uniforms replayed.  Units whose cached tensors the fixture holds run on those and must take the
    reference's first Adam step and end with its hard rounding; the others run on the product's own caches (loose bound).
    Includes the reference's checkpoint behaviour of QuantAttentionBlock (backward on re-drawn masks, no gradient from the
    per-module loss: quant_block.py:180-182, util.py:117-148)."""
    from qdiff import Change_LDM_model_attnblock, recon_block_Qmodel
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.quant_layer import UniformAffineQuantizer
    from qdiff.quant_block import QuantAttentionBlock
    from edadm.state import load_quant_state
    import edadm.recon as recon
    import qdiff.data_utils as du
    g = golden("g18_church_driver")
    qnn, ld = _church_ld(golden)
    cali = (_cuda(g["tdac/calib_data"]), _cuda(g["tdac/t"]), _cuda(g["tdac/index"]))
    qnn.model.split_shortcut = True
    with torch.no_grad():
        qnn(cali[0][:2], cali[1][:2])                    # creates the split quantizers of the skip convolutions
    n = load_quant_state(qnn, {k: g[k] for k in g.files if k.startswith("init/qp/")}, prefix="init/qp/")
    assert n == len([k for k in g.files if k.startswith("init/qp/") and k.endswith("/delta")])
    Change_LDM_model_attnblock(qnn, dict(AQ8))
    assert sum(isinstance(m, QuantAttentionBlock) for m in qnn.modules()) == 7
    iters = int(g["iters"])
    kwargs = dict(cali_data=cali[:-1], iters=iters, act_quant=True, asym=True, opt_mode='mse', lr_a=1e-4, lr_w=5e-2, p=2.0,
                  weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=0.5, add_loss=1.0, recon_w=True,
                  recon_a=True, keep_gpu=False)
    rep = _uniforms.Replay()
    cur = {"name": None, "phase": "iter"}
    names = {m: nme for nme, m in qnn.named_modules()}
    for name, m in qnn.named_modules():
        if isinstance(m, UniformAffineQuantizer) and m.leaf_param:
            m.injected_uniform = (lambda nm: lambda xx: torch.from_numpy(rep.draw_calls(nm, cur["phase"], xx.shape, recon.STATE["batched"] and cur["phase"] == "iter")).to(xx.device))(name)
    recon.INJECT_MIX_UNIFORM = lambda xx: torch.from_numpy(rep.draw("input_mix:" + cur["name"], "iter", xx.shape)).to(xx.device)
    order, traj, idx_log = [], {}, {}
    import qdiff.block_recon as brm
    import qdiff.layer_recon as lrm
    orig_save, orig_launch, orig_sample, orig_rec = du.save_inp_oup_data, recon.FusedAdam.launch, random.sample, recon.reconstruct

    def save_fn(model, unit, cali_data, asym, act_quant, batch_size=32, input_prob=True, keep_gpu=True):
        k = "cache/%s/" % cur["name"]
        if k + "out_fp" in g.files:
            if bool(g[k + "resblock"]):
                return True, ([_cuda(g[k + "inp_q"]), _cuda(g[k + "temb_q"])], [_cuda(g[k + "inp_fp"]), _cuda(g[k + "temb_fp"])]), \
                    _cuda(g[k + "out_fp"])
            return False, (_cuda(g[k + "inp_q"]), _cuda(g[k + "inp_fp"])), _cuda(g[k + "out_fp"])
        cur["phase"] = "cache"
        try:
            return orig_save(model, unit, cali_data, asym, act_quant, batch_size=batch_size, input_prob=input_prob, keep_gpu=keep_gpu)
        finally:
            cur["phase"] = "iter"

    def rec(model, unit, cali_data, **kw):
        cur["name"] = names[unit]
        order.append("%s:%s:%s" % ("block" if kw["is_block"] else "layer", names[unit], type(unit).__name__))
        return orig_rec(model, unit, cali_data, **kw)

    def launch(self):
        orig_launch(self)
        key = "%s/%s" % (cur["name"], "a" if self.params[0].numel() == 1 else "w")
        traj.setdefault(key, []).append(self.flat.detach().cpu().clone())

    def sample(pop, k):
        r = orig_sample(pop, k)
        idx_log.setdefault(cur["name"], []).append(list(r))
        return r

    du.save_inp_oup_data, recon.FusedAdam.launch, random.sample = save_fn, launch, sample
    brm.reconstruct = lrm.reconstruct = rec
    try:
        random.seed(1818)
        qnn.set_quant_state(True, True)
        recon_block_Qmodel(SimpleNamespace(), qnn, cali, kwargs).recon()
    finally:
        du.save_inp_oup_data, recon.FusedAdam.launch, random.sample = orig_save, orig_launch, orig_sample
        brm.reconstruct = lrm.reconstruct = orig_rec
        recon.INJECT_MIX_UNIFORM = None
    assert order == list(g["order"])
    for k in idx_log:
        assert np.array_equal(np.asarray(idx_log[k]), g["idx/" + k]), k
    ref_log = sorted(l for l in g["rand/log"] if "|iter|" in l)
    got_log = sorted("%s|%s|%d|%s" % (o, p, c, "x".join(map(str, s))) for o, p, c, s in rep.log if p == "iter")
    assert got_log == ref_log, (sorted(set(got_log) - set(ref_log))[:6], sorted(set(ref_log) - set(got_log))[:6])
    cached_units = sorted({k.split("/")[1] for k in g.files if k.startswith("cache/")}, key=len, reverse=True)
    for u in [o.split(":")[1] for o in order]:
        cached = u in cached_units
        d0 = np.abs(traj[u + "/w"][0].numpy() - g["traj/%s/w" % u])
        ra, ra0 = 0.0, 0.0
        if "traj/%s/a" % u in g.files:
            ref_a = g["traj/%s/a" % u]
            da = np.abs(torch.stack(traj[u + "/a"]).numpy() - ref_a) / 1e-4            # in Adam steps of lr_a
            ra, ra0 = float(da.max()), float(da[0].max())
        print(u, "reference caches" if cached else "own caches", "first step: frac>1e-2 %.5f max %.3g | step sizes: first step off "
              "by %.3g lr_a, worst over %d steps %.3g lr_a" % ((d0 > 1e-2).mean(), d0.max(), ra0, iters, ra))
        if cached:
            # +-lr_w by the sign of the first gradient: the reference's step except where that gradient is rounding noise
            # step sizes: the first Adam step identical; later ones may part where lr_a is a fifth of the step size itself (the
            # softmax-probability quantizer of the 64-key attention: delta 4e-4, lr_a 1e-4 -- every step moves a fifth of the
            # codes) -- bounded by one step of lr_a over the unit
            assert (d0 > 1e-2).mean() < 2e-3 and ra0 < 2e-2 and ra < 1.0, (u, (d0 > 1e-2).mean(), ra0, ra)
        else:
            assert (d0 > 1e-2).mean() < 0.2, (u, (d0 > 1e-2).mean())
    agree, total, bad = 0, 0, []
    for name, m in qnn.named_modules():
        if isinstance(m, AdaRoundQuantizer):
            ref_alpha = g["final/alpha/" + name]
            got = m.alpha.detach().cpu().numpy()
            dis = (got >= 0) != (ref_alpha >= 0)
            agree += int((~dis).sum())
            total += got.size
            if any(name.startswith(u + ".") for u in cached_units):
                bad += [(name, float(r), float(o)) for r, o in zip(ref_alpha[dis], got[dis])]
    print("final hard rounding: %d of %d agree (%.4f %%); on the reference's caches: %d disagreements %s" % (
        agree, total, 100.0 * agree / total, len(bad), bad[:8]))
    assert total == sum(g[k].size for k in g.files if k.startswith("final/alpha/"))
    # measured (round 4): 99.808 % over the whole walk (units after the first run on the product's OWN quantised prefix, which carries
    # the near-zero alphas of the units before), 1 disagreement on the units fed the reference's caches; gates at 2x
    assert agree / total > 0.996
    assert all(abs(r) < 1e-2 and abs(o) < 1e-2 for _, r, o in bad) and len(bad) <= 2, bad
    assert qnn.block_count == int(g["block_count"])
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        out = qnn(cali[0][:8], cali[1][:8]).cpu().numpy()
    ref = g["final/out_q"]
    err = np.abs(out - ref) / np.abs(ref).max()
    print("final quantised output vs reference: max %.3f mean %.4f of range" % (err.max(), err.mean()))
    assert err.max() < 0.25 and err.mean() < 0.04


@pytest.mark.parametrize("which", ["cifar", "bedroom", "coco"])
def test_tdac_generator_values_other_configs(golden, which):
    """VALUES of the remaining TDAC generators against the reference's own runs (fixture G19; Church is in G18, ImageNet in G17):
    TDAC_cifar_calib_data_generator (calibration.py:12-155: one trajectory batch serves every calibration batch),
    TDAC_bedroom_… (:156-262, `> 0` fix-up), TDAC_coco_… (:502-638: PLMS + classifier-free guidance, t_next).  Start noise and
    permutation injected; timesteps / indices exact, latents to trajectory accuracy."""
    import scripts.calibration as cal
    from edadm.latent import LatentDiffusionLite
    from qdiff import QuantModel
    from helpers import build_cifar, build_ldm, build_ldm_formula
    g = golden("g19_tdac_others")
    P = which + "/"
    N, nb = int(g[P + "N"]), int(g[P + "nb"])
    dev = torch.device("cuda")
    if which == "cifar":
        base = golden("g13_cifar_unet")
        qnn = QuantModel(build_cifar(base), WQ4, dict(AQ8), sm_abit=8).cuda().eval()
        qnn.set_quant_state(False, False)
        diffusion = SimpleNamespace(seq=[int(v) for v in g[P + "seq"]], betas=torch.linspace(1e-4, 2e-2, 1000).cuda(),
                                    args=SimpleNamespace(eta=0.0))
        draws = iter([torch.as_tensor(g[P + "x_T"])])
        run = lambda: cal.TDAC_cifar_calib_data_generator(qnn.model, qnn.model.config, float(g[P + "lamda"]), N, nb, dev, diffusion, True)
    elif which == "bedroom":
        base = golden("g13_ldm_church")
        qnn = QuantModel(build_ldm(base), WQ4, dict(AQ8), sm_abit=8).cuda().eval()
        qnn.set_quant_state(False, False)
        ld = LatentDiffusionLite(qnn, timesteps=1000, linear_start=0.0015, linear_end=0.0195, conditioning_key=None).cuda()
        S = int(g[P + "S"])
        args = SimpleNamespace(custom_steps=S, eta=0.0, lamda=float(g[P + "lamda"]))
        draws = iter(torch.as_tensor(g[P + "x_T"]))
        run = lambda: cal.TDAC_bedroom_calib_data_generator(ld, args, N, nb, dev, S)
    else:
        base = golden("g13_ldm_sd")
        qnn = QuantModel(build_ldm_formula(base), WQ4, dict(AQ8), sm_abit=8).cuda().eval()
        qnn.set_quant_state(False, False)
        qnn.set_grad_ckpt(False)
        ld = LatentDiffusionLite(qnn, timesteps=1000, linear_start=0.00085, linear_end=0.012, conditioning_key="crossattn").cuda()
        table = _cuda(g[P + "table"])
        prompts = ["p%d" % i for i in range(N)]
        look = {"": table[0], **{p: table[i + 1] for i, p in enumerate(prompts)}}
        ld.get_learned_conditioning = lambda ps: torch.stack([look[p] for p in ps])
        S = int(g[P + "S"])
        args = SimpleNamespace(custom_steps=S, scale=float(g[P + "scale"]), ddim_eta=0.0, plms=True, C=4, H=64, W=64, f=8,
                               list_prompts=prompts, lamda=float(g[P + "lamda"]))
        draws = iter(torch.as_tensor(g[P + "x_T"]))
        run = lambda: cal.TDAC_coco_calib_data_generator(ld, args, N, nb, dev, S)
    orig_randn, orig_perm = torch.randn, torch.randperm
    torch.randn = lambda *a, **k: next(draws).cuda()
    torch.randperm = lambda n, **k: torch.as_tensor(g[P + "perm"])
    try:
        out = run()
    finally:
        torch.randn, torch.randperm = orig_randn, orig_perm
    names = {"cifar": ("calib_data", "t", "cls"), "bedroom": ("calib_data", "t", "index"),
             "coco": ("calib_data", "t", "index", "cond", "uncond", "t_next")}[which]
    assert len(out) == len(names)
    for nme, got in zip(names, out):
        ref = g[P + nme]
        got = got.detach().cpu().numpy()
        assert got.shape == ref.shape, (nme, got.shape, ref.shape)
        if ref.dtype.kind in "iu":
            np.testing.assert_array_equal(got, ref, err_msg=nme)
        else:
            err = np.abs(got - ref).max() / np.abs(ref).max()
            print(which, nme, "vs the reference generator: max %.2e of range" % err)
            assert err <= 1e-4, (nme, err)


def test_scale_method_max_bit_exact(golden):
    """scale_method='max' -- the reference constructor's default (quant_layer.py:48,278-330) -- through the product quantizer on
    the device: per-channel weight and per-tensor activation step sizes / zero points bit-exact against fixture G1b, and the
    fake-quant forward (K1) that follows."""
    from qdiff.quant_layer import UniformAffineQuantizer
    import torch.nn as nn
    g = golden("g1b_max_init")
    n = 0
    for k in g.files:
        if not k.endswith("/delta"):
            continue
        key, parts = k[:-6], k.split("/")
        x = _cuda(g[("w/" if parts[0].startswith(("conv", "lin")) else "x/") + parts[0]])
        if parts[1].startswith("b"):
            q = UniformAffineQuantizer(n_bits=int(parts[1][1:]), symmetric=parts[2] == "sym", channel_wise=True, scale_method="max")
        else:
            q = UniformAffineQuantizer(n_bits=8, symmetric=parts[1] == "sym", channel_wise=False, scale_method="max",
                                       leaf_param=True, always_zero=parts[2] == "az")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = q(x)
        assert isinstance(q.delta, nn.Parameter) == q.leaf_param
        np.testing.assert_array_equal(q.delta.detach().cpu().numpy().reshape(-1), g[key + "/delta"].reshape(-1), err_msg=key)
        np.testing.assert_array_equal(q.zero_point.cpu().numpy().reshape(-1), g[key + "/zero_point"].reshape(-1), err_msg=key)
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[key + "/out"], rtol=1e-6, atol=1e-7, err_msg=key)
        n += 1
    assert n == 36
    assert UniformAffineQuantizer().scale_method == "max"               # the default constructs and runs
    UniformAffineQuantizer()(torch.randn(4, 4, device="cuda"))


def test_deferred_device_status_and_rejected_misaligned_pair_output():
    """include/edadm.h edadm_device_status: after healthy launches of the persistent GEMM (the kernel whose hand-off wait can
    give up) the deferred error word is clean; the GEGLU-pair output form refuses an output pointer its vector epilogue cannot
    serve (-EINVAL) instead of falling into the element-wise form that does not know the pair layout."""
    from edadm import ops, lib
    g = torch.Generator().manual_seed(5)
    M, N, K = 102400, 3072, 384                                   # the LDM-4 GEGLU projection: persistent kernel
    A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    W = torch.randint(-8, 9, (N, K), generator=g, dtype=torch.int8).cuda()
    scale, bias = (torch.rand(N, generator=g) * 1e-3 + 1e-4).cuda(), torch.randn(N, generator=g).cuda()
    qp = ops.qp_tensor([(0.041, 123.0, 255.0)], "cuda")
    out = ops.qgemm_i8_q(A, W, M, N, K, scale, bias, 3, qp)
    assert out.shape == (M, N // 2)
    ops.device_status()                                            # synchronises; raises on a recorded hand-off timeout
    assert lib.load().edadm_device_status(0, None) == 0
    buf = torch.empty(M * (N // 2) + 16, dtype=torch.int8, device="cuda")
    import ctypes
    rc = lib.load().edadm_qgemm_i8_q(A.data_ptr(), K, W.data_ptr(), K, M, N, K, None, scale.data_ptr(), bias.data_ptr(), None, 1,
                                     None, 0, buf.data_ptr() + 1, N // 2, 3, qp.data_ptr(), None)
    assert rc == -22


@pytest.mark.parametrize("B,heads,d,Nq,Nk", [(2, 8, 8, 16, 16), (2, 8, 8, 16, 77), (2, 2, 16, 16, 7), (2, 8, 40, 200, 77), (2, 8, 24, 96, 96),
                                             (1, 8, 40, 1024, 1024), (2, 8, 80, 256, 256), (1, 4, 160, 64, 64), (2, 1, 32, 130, 130),
                                             (1, 1, 384, 1024, 1024), (2, 1, 384, 256, 256), (3, 2, 384, 200, 64),
                                             (1, 1, 576, 256, 256), (3, 1, 576, 200, 256), (1, 1, 960, 64, 64), (2, 2, 960, 100, 64)])
def test_fused_attention_against_torch_on_the_same_codes(B, heads, d, Nq, Nk):
    """K6f (csrc/attn.hip, edadm_attention_fused_f16): quantise -> Q K^T -> softmax -> 8-bit probability codes -> P V in one kernel,
    no score matrix in memory (quant_block.py:204-235 / :119-162 / :398-451), against torch on the same integer codes.  The
    products are exact; what may differ is a probability code that sits on a rounding boundary (fp32 row-sum order, last bit of
    the exponential): outputs equal except in a small fraction of elements, each off by at most two codes' worth; also the
    legacy (q|k|v)-per-head operand layout, the int8-operand output form, and agreement with the three-kernel path."""
    from edadm import ops
    from edadm.engine import Engine
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 1000 + heads * 100 + d + Nk)
    hd = heads * d
    q, k, v = (torch.randn(B * n, hd, generator=g).to(dev) for n in (Nq, Nk, Nk))
    mkq = lambda delta, zp: SimpleNamespace(delta=torch.tensor(delta, device=dev), zero_point=torch.tensor(float(zp), device=dev), n_levels=256)
    aq, ak, av, aw = mkq(0.03, 128), mkq(0.031, 127), mkq(0.029, 128), mkq(1 / 255.0, 0)
    scale = d ** -0.5

    def run(fused, **kw):
        eng = Engine.__new__(Engine)
        eng.dev, eng._attn_cache, eng.fused_attention = dev, {}, fused
        return eng.attention(q, k, v, B, Nq, Nk, heads, d, aq, ak, av, aw, scale, **kw)

    out = run(True)
    codes = lambda x, qz: torch.clamp(torch.round(x / qz.delta) + qz.zero_point, 0, 255) - qz.zero_point
    sp = lambda t, n: t.reshape(B, n, heads, d).permute(0, 2, 1, 3)
    cq, ck, cv = codes(q, aq), codes(k, ak), codes(v, av)
    s = torch.einsum("bhid,bhjd->bhij", sp(cq, Nq).double(), sp(ck, Nk).double()) * float(aq.delta * ak.delta) * scale
    cp = codes(torch.softmax(s.float(), -1), aw)
    ref = (torch.einsum("bhij,bhjd->bhid", cp.double(), sp(cv, Nk).double()) * float(aw.delta * av.delta)).permute(0, 2, 1, 3).reshape(B * Nq, hd).float()
    one_code = 128.0 * float(aw.delta * av.delta)                     # one probability code x the largest value code
    for name, other in (("torch on the codes", ref), ("three-kernel path", run(False))):
        diff = (out - other).abs()
        frac = float((diff > 1e-6).float().mean())
        print("B=%d heads=%d d=%d Nq=%d Nk=%d vs %s: %.4f %% of outputs differ, max %.2f codes" % (B, heads, d, Nq, Nk, name, 100 * frac,
                                                                                                  float(diff.max()) / one_code))
        assert frac < 5e-3 and float(diff.max()) <= 2.0 * one_code + 1e-6
    # the consumer's int8 operand straight from the epilogue = quantising the fp32 output
    oqp = ops.qp_tensor([(0.037, 131.0, 255.0)], dev)
    got = run(True, out_qp=oqp)
    want = ops.quant_i8(out, oqp)
    assert got.dtype == torch.int8 and float((got != want).float().mean()) < 1e-3
    # legacy layout: one [rows][heads x (q|k|v) x d] tensor (openaimodel.py:390-393), Nq == Nk
    if Nq == Nk:
        qkv = torch.stack([q.reshape(-1, heads, d), k.reshape(-1, heads, d), v.reshape(-1, heads, d)], 2).reshape(-1, 3 * hd).contiguous()
        eng = Engine.__new__(Engine)
        eng.dev, eng._attn_cache, eng.fused_attention = dev, {}, True
        leg = eng.attention(qkv, qkv, qkv, B, Nq, Nk, heads, d, aq, ak, av, aw, scale, qcols=[h * 3 * d for h in range(heads)],
                            kcols=[h * 3 * d + d for h in range(heads)], vcols=[h * 3 * d + 2 * d for h in range(heads)])
        assert torch.equal(leg, out)


@pytest.mark.parametrize("M,N,K", [(100, 384, 768), (2000, 768, 192), (2000, 1920, 768), (64, 64, 32), (37, 200, 96), (2048, 576, 512),
                                   (8, 1280, 1280), (8, 320, 320), (50, 96, 1248), (1, 40, 64)])
def test_w4_gemm_reads_nibbles_bit_identical_to_int8(M, N, K):
    """K4w (csrc/w4.hip, edadm_qgemm_w4): the few-row layers read their 4-bit weights as packed nibbles (the frozen file's own
    format, edadm_pack_w4) and expand them in registers; integer arithmetic -> the bits of edadm_qgemm_i8 on the unpacked int8
    copy, with bias, per-image row add and residual (quant_layer.py:406-437 at inference)."""
    from edadm import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8).cuda()
    codes = torch.randint(0, 16, (N, K), generator=g)
    zp = torch.randint(0, 16, (N,), generator=g).float()
    W = (codes - zp[:, None].long()).to(torch.int8).cuda()
    zp = zp.cuda()
    scale, bias = (torch.rand(N, generator=g) * 1e-3 + 1e-4).cuda(), torch.randn(N, generator=g).cuda()
    rpb = 50 if M % 50 == 0 else M
    rowadd, res = torch.randn(M // rpb, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    packed = ops.pack_w4(W, zp)
    assert torch.equal(ops.unpack_w4(packed, zp, N, K), W)
    cases = [{}, dict(residual=res)]
    if rpb >= 16:                                   # the int8 kernel stages at most BM / 16 + 1 row-add rows per tile
        cases += [dict(rowadd=rowadd, rows_per_batch=rpb), dict(rowadd=rowadd, rows_per_batch=rpb, residual=res)]
    for kw in cases:
        want = ops.qgemm_i8(A, W, M, N, K, scale, bias, torch.empty(M, N, device="cuda"), **kw)
        got = ops.qgemm_w4(A, packed, zp, M, N, K, scale, bias, torch.full((M, N), float("nan"), device="cuda"), **kw)
        assert torch.equal(got, want), (kw.keys(), float((got - want).abs().max()))


def test_engine_few_row_layers_take_the_nibble_path(golden):
    """The frozen executor routes its few-row dense 4-bit layers (time embedding, one-token context branches) through K4w and the
    network output keeps its bits; a frozen file's packed arrays feed the kernel directly after load_frozen()."""
    from helpers import build_ldm, quantize_like_reference
    g = golden("g13_ldm_imagenet")
    qnn, (x, t, ctx), _ = quantize_like_reference(build_ldm(g), g, "ldm")
    qnn.set_quant_state(True, True)
    eng = qnn.freeze()
    eng.prof = []
    with torch.no_grad():
        out_w4 = eng(x, t, ctx).clone()
    modes = [p[0] for p in eng.prof]
    eng.prof = None
    assert modes.count("w4") >= 10, modes
    eng.w4_gemm_max_rows = 0
    with torch.no_grad():
        out_i8 = eng(x, t, ctx)
    assert torch.equal(out_w4, out_i8)
    eng.w4_gemm_max_rows = 2048
    eng.load_frozen(eng.export_frozen())
    with torch.no_grad():
        assert torch.equal(eng(x, t, ctx), out_i8)


def _run_harness(mod, common, calib_args, sample_args, tmp_path, capsys, batches):
    import json
    out = str(tmp_path / "calib")
    mod.main(["calibrate", "--out", out] + calib_args + common)
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["job"] == "calibrate" and line["units"] > 5 and line["frozen_bytes"] > 0 and line["reconstruction_s"] > 0
    saves = [str(tmp_path / "a"), str(tmp_path / "b")]
    for save in saves:
        mod.main(["sample", "--state", out, "--save", save] + sample_args + common)
        line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
        assert line["job"] == "sample" and line["ranks"] == 1 and line["batches_this_rank"] == batches
    for i in range(batches):
        a, b = np.load("%s/batch_%06d.npy" % (saves[0], i)), np.load("%s/batch_%06d.npy" % (saves[1], i))
        assert np.isfinite(a).all() and np.array_equal(a, b), i
    assert not np.array_equal(np.load(saves[0] + "/batch_000000.npy"), np.load(saves[0] + "/batch_000001.npy"))
    return line


def test_task_harness_cifar_ddim(golden, tmp_path, capsys):
    """scripts/sample_diffusion_ddim.py (the flow of the reference's script of that name, :265-323; BASELINE configs 1 / 2) on a
    fixture-sized DDPM UNet: TDAC_cifar set -> scale init -> recon_block_Qmodel -> state + frozen model -> load -> DDIM sampling on
    the int8 executor, twice with the same result."""
    import json
    from scripts import sample_diffusion_ddim as H
    base = golden("g13_cifar_unet")
    model = dict(type="simple", in_channels=3, out_ch=3, ch=int(base["cfg/ch"]), ch_mult=[int(v) for v in base["cfg/ch_mult"]],
                 num_res_blocks=int(base["cfg/nres"]), attn_resolutions=[int(v) for v in base["cfg/attn"]], dropout=0.0,
                 resamp_with_conv=True, image_size=int(base["cfg/res"]))
    common = ["--model", json.dumps(model), "--timesteps", "20"]
    _run_harness(H, common, ["--calib_num_samples", "32", "--batch_samples", "32", "--iters", "2"],
                 ["--max_images", "16", "--n_batch", "8"], tmp_path, capsys, 2)


def test_task_harness_church_ldm(golden, tmp_path, capsys):
    """scripts/sample_diffusion_ldm_church.py (reference :256-311; BASELINE config 3): TDAC_church -> set_*_quantize_params_LDM ->
    Change_LDM_model_attnblock -> unconditional walk -> state -> load (attention blocks wrapped again) -> unguided DDIM sampling."""
    import json
    from scripts import sample_diffusion_ldm_church as H
    base = golden("g13_ldm_church")
    kw = {k[4:]: (base[k].tolist() if base[k].ndim else base[k].item()) for k in base.files if k.startswith("cfg/")}
    common = ["--unet", json.dumps(kw), "--custom_steps", "20"]
    _run_harness(H, common, ["--calib_num_samples", "32", "--batch_samples", "8", "--iters", "2"],
                 ["--n_samples", "8", "--batch_size", "4"], tmp_path, capsys, 2)


def test_task_harness_bedroom_ldm(golden, tmp_path, capsys):
    """scripts/sample_diffusion_ldm_bedroom.py (reference :257-316): the Church flow over TDAC_bedroom_calib_data_generator with the
    bedroom task table (eta 1 DDIM: the stochastic update's noise comes from the per-batch generator, so two runs agree)."""
    import json
    from scripts import sample_diffusion_ldm_bedroom as H
    from scripts import sample_diffusion_ldm_church as C
    base = golden("g13_ldm_church")
    kw = {k[4:]: (base[k].tolist() if base[k].ndim else base[k].item()) for k in base.files if k.startswith("cfg/")}
    common = ["--unet", json.dumps(kw), "--custom_steps", "20"]
    assert H.parser().parse_args(["calibrate"]).eta == 1.0 and H.parser().parse_args(["calibrate"]).custom_steps == 200
    _run_harness(H, common, ["--calib_num_samples", "32", "--batch_samples", "8", "--iters", "2"],
                 ["--n_samples", "8", "--batch_size", "4", "--eta", "0.0"], tmp_path, capsys, 2)
    assert C.TASK["tdac"] == "TDAC_church_calib_data_generator"          # the task table is restored after the run


def test_task_harness_txt2img_sd(golden, tmp_path, capsys):
    """scripts/sample_txt2img.py (reference :154-283; BASELINE config 5) on a Stable-Diffusion-shaped fixture UNet (8 heads, 77-token
    context): TDAC_coco through the PLMS sampler with guidance -> set_*_quantize_params_Stable -> conditional walk at batch 2 ->
    state -> load -> PLMS + CFG sampling on the int8 executor (fused attention K6f on the 77-key cross-attention)."""
    import json
    from scripts import sample_txt2img as H
    base = golden("g13_ldm_sd")
    kw = {k[4:]: (base[k].tolist() if base[k].ndim else base[k].item()) for k in base.files if k.startswith("cfg/")}
    common = ["--unet", json.dumps(kw), "--custom_steps", "30", "--H", "64", "--W", "64"]
    _run_harness(H, common, ["--calib_num_samples", "8", "--batch_samples", "2", "--iters", "2"],
                 ["--n_samples", "4", "--n_batch", "2"], tmp_path, capsys, 2)


@pytest.mark.parametrize("B,heads,Nq,Nk,zq,zk", [(1, 1, 1024, 1024, 128, 128), (2, 1, 256, 256, 127, 128), (2, 2, 200, 128, 125, 131), (1, 1, 64, 64, 128, 127)])
def test_wide_head_attention_with_int8_scores_against_torch_on_the_same_codes(B, heads, Nq, Nk, zq, zk):
    """K6w, int8 score form (csrc/attn.hip k_attn_wide16_i8, edadm_attention_fused_i8qk): q and k as int8 operands (code - 128), the
    (128 - z_q) sum_d k8 correction per key, key-independent terms dropped (they cancel in the softmax), v as f16 codes -- against
    torch on the same integer codes (quant_block.py:204-235) and against the f16 form of the same kernel family; any zero points."""
    from edadm import ops
    dev = torch.device("cuda")
    d = 384
    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk + zq)
    hd = heads * d
    cq = torch.randint(0, 256, (B * Nq, hd), generator=g).to(dev)
    ck = torch.randint(0, 256, (B * Nk, hd), generator=g).to(dev)
    cv = torch.randint(0, 256, (B * Nk, hd), generator=g).to(dev)
    zv, dq_, dk_, dv_, dw_ = 128, 0.03, 0.031, 0.029, 1.0 / 255.0
    scale = d ** -0.5
    q8, k8 = (cq - 128).to(torch.int8), (ck - 128).to(torch.int8)
    vh = (cv - zv).to(torch.float16)
    pqp = ops.qp_tensor([(torch.tensor(dw_), torch.tensor(0.0), 255)], dev)
    alpha = dq_ * dk_ * scale * 0.4                       # random codes: logits of a few units -- several keys per query keep a non-zero code
    out = ops.attention_fused_i8qk(q8, k8, vh, B, heads, Nq, Nk, d, alpha, float(zq), pqp, dw_ * dv_)
    sp = lambda t, n: t.reshape(B, n, heads, d).permute(0, 2, 1, 3).double()
    s = torch.einsum("bhid,bhjd->bhij", sp(cq - zq, Nq), sp(ck - zk, Nk)) * alpha
    p = torch.softmax(s.float(), -1)
    cp = torch.clamp(torch.round(p / dw_), 0, 255)
    ref = (torch.einsum("bhij,bhjd->bhid", cp.double(), sp(cv - zv, Nk)) * (dw_ * dv_)).permute(0, 2, 1, 3).reshape(B * Nq, hd).float()
    one_code = 128.0 * dw_ * dv_
    diff = (out - ref).abs()
    frac = float((diff > 1e-6).float().mean())
    print("int8 scores B=%d heads=%d Nq=%d Nk=%d zq=%d zk=%d vs torch on the codes: %.4f %% of outputs differ, max %.2f codes; mean |out| %.3f"
          % (B, heads, Nq, Nk, zq, zk, 100 * frac, float(diff.max()) / one_code, float(ref.abs().mean())))
    assert float(ref.abs().max()) > 10 * one_code                     # the probabilities are not all rounded away
    assert frac < 5e-3 and float(diff.max()) <= 2.0 * one_code + 1e-6
    # the f16 form of the kernel on the same codes (both operands minus their own zero points)
    qh, kh = (cq - zq).to(torch.float16), (ck - zk).to(torch.float16)
    f16 = ops.attention_fused(qh, kh, vh, B, heads, Nq, Nk, d, alpha, pqp, dw_ * dv_)
    d2 = (out - f16).abs()
    assert float((d2 > 1e-6).float().mean()) < 5e-3 and float(d2.max()) <= 2.0 * one_code + 1e-6
    oqp = ops.qp_tensor([(torch.tensor(0.037), torch.tensor(131.0), 255)], dev)
    got = ops.attention_fused_i8qk(q8, k8, vh, B, heads, Nq, Nk, d, alpha, float(zq), pqp, dw_ * dv_, out_qp=oqp)
    want = ops.quant_i8(out, oqp)
    assert got.dtype == torch.int8 and float((got != want).float().mean()) < 1e-3
