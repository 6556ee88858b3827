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
                rep.draw(nm, "iter", xx.shape).transpose(0, 2, 1).copy() if tr else rep.draw(nm, "iter", xx.shape)).to(xx.device))(name, tr)
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
    assert all(abs(r) < 1e-2 and abs(o) < 1e-2 for _, _, r, o in bad) and len(bad) <= 27, bad


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
