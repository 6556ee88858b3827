"""Round-3 oracle pins (CPU): the reconstruction loop with the SHIPPED stochastic setting (prob = input_prob = 0.5,
sample_diffusion_ldm_imagenet.py:144,185) against the reference's own trajectories -- fixture G8b, whose uniforms
(block_recon.py:141-145, quant_layer.py:271-275) come from tests/golden/_uniforms.py through a patched torch.rand_like --
and the index-space variant G8c (reference scales LOADED, reference caches injected, hard rounding must agree)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from oracle import qdiff_oracle as O
from test_oracle_nets import ToyNet, sub_sd, close, T, WQ4, AQ8

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import _uniforms  # noqa: E402

UNITS = (("conv_in", "layer"), ("temb_lin", "layer"), ("rb", "block"), ("at", "block"), ("conv_out", "layer"))


def check_uniform_generator(g):
    """the uniforms regenerated here are the ones the reference consumed (first / last value and sum of every call)"""
    for line, chk in zip(g["rand/log"], g["rand/check"]):
        owner, phase, c, shp = line.split("|")
        u = _uniforms.uniform(owner, phase, int(c), [int(s) for s in shp.split("x")])
        assert float(u.reshape(-1)[0]) == chk[0] and float(u.reshape(-1)[-1]) == chk[1]
        assert abs(float(u.astype(np.float64).sum()) - chk[2]) < 1e-9


def golden_caches(g, name):
    k = "cache/%s/" % name
    two = bool(g[k + "resblock"])
    iq = (T(g[k + "inp_q"]), T(g[k + "temb_q"])) if two else (T(g[k + "inp_q"]),)
    ifp = (T(g[k + "inp_fp"]), T(g[k + "temb_fp"])) if two else (T(g[k + "inp_fp"]),)
    return two, iq, ifp, T(g[k + "out_fp"])


def load_init_scales(net, g):
    n = 0
    for q in net.all_quantizers():
        k = "init/qp/model." + q.name
        if k + "/delta" in g.files:
            q.delta = T(g[k + "/delta"]).float()
            q.zero_point = T(g[k + "/zero_point"]).float()
            q.bitwidth_refactor(int(g[k + "/n_bits"]))
            q.inited = True
            n += 1
    return n


@pytest.mark.parametrize("fixture", ["g8c_recon_caches", "g8b_recon_masks"])
def test_recon_loop_with_reference_scales_caches_and_masks(golden, fixture):
    g = golden(fixture)
    prob, input_prob, iters = float(g["prob"]), float(g["input_prob"]), int(g["iters"])
    check_uniform_generator(g)
    aq = dict(AQ8)
    aq["prob"] = prob
    net = ToyNet(sub_sd(g, "sd/"), WQ4, aq)
    x, t = T(g["x"]), T(g["t"])
    with torch.no_grad():
        net(x, t)
    assert load_init_scales(net, g) == len([k for k in g.files if k.startswith("init/qp/") and k.endswith("/delta")])
    rep = _uniforms.Replay()
    for q in net.all_quantizers():
        if isinstance(q, O.OQ):
            q.mask_fn = (lambda name: lambda xx: torch.from_numpy(rep.draw("model." + name, "iter", xx.shape)))(q.name)
    random.seed(8080)
    for name, kind in UNITS:
        unit = getattr(net, name)
        # the oracle's own caches equal the reference's (the unit's own training-mode draws during the reference's caching
        # pass -- phase "cache" of the log -- touch nothing that is kept: the hook stores the unit's INPUT and the FP output);
        # then the loop runs on the reference's tensors
        net.set_quant_state(True, True)
        two, iq, ifp, ofp = O.save_inp_oup_data(net, unit, (x, t), True, 32)
        rtwo, riq, rifp, rofp = golden_caches(g, name)
        assert two == rtwo
        for a, b in zip(ifp + (ofp,), rifp + (rofp,)):
            close(a, b, rtol=1e-5, atol=1e-6)
        for a, b in zip(iq, riq):
            dq = (a - b).abs()
            assert float(dq.median()) < 3e-3      # exact until an upstream unit ends with one of the near-zero alphas below
            print(name, "own cache vs reference: median %.2e frac>1e-3 %.4f max %.3g" % (float(dq.median()), float((dq > 1e-3).float().mean()), float(dq.max())))
        tw, ta = [], []
        O.reconstruct_unit(net, unit, kind, cali=(x, t), iters=iters, act_quant=True, lr_a=1e-3, lr_w=5e-2, p=2.0, batch_size=16,
                           input_prob=input_prob, add_loss=0.8, recon_w=True, recon_a=True, cache_batch=32,
                           caches=(rtwo, riq, rifp, rofp),
                           rand_fn=lambda xx, n=name: torch.from_numpy(rep.draw("input_mix:" + n, "iter", xx.shape)),
                           trace=lambda it, wp, ap, l: (tw.append(torch.cat([p.detach().flatten() for p in wp]).clone()),
                                                        ta.append(torch.cat([p.detach().flatten() for p in ap]).clone())))
        ref_w, ref_a = g["traj/%s/w" % name], g["traj/%s/a" % name]
        got_w, got_a = torch.stack(tw).numpy(), torch.stack(ta).numpy()
        dw = np.abs(got_w - ref_w)
        print(fixture, name, "alpha: median %.2e frac>1e-2 %.5f max %.3g | delta max rel %.3g" % (
            np.median(dw), (dw > 1e-2).mean(), dw.max(), (np.abs(got_a - ref_a) / np.abs(ref_a)).max()))
        assert np.median(dw) < 5e-4, name
        assert (dw > 1e-2).mean() < 5e-3 and dw.max() < 2 * 5e-2, (name, (dw > 1e-2).mean(), dw.max())
        # the attention block's softmax-probability step size (4e-3) is driven THROUGH zero by the fixture's lr_a = 1e-3
        # (x / delta blows up there in the reference as well): half an Adam step of slack for that unit
        np.testing.assert_allclose(got_a, ref_a, rtol=5e-3, atol=6e-4 if name == "at" else 1e-6)
    # every draw of the iterations was consumed in the reference's per-quantizer order and shape
    ref_log = sorted(l for l in g["rand/log"] if "|iter|" in l)
    got_log = sorted("%s|%s|%d|%s" % (o, p, c, "x".join(map(str, s))) for o, p, c, s in rep.log)
    assert got_log == ref_log
    # index space: the final hard rounding of every weight
    bad = []
    for l in net.all_layers():
        key = "model.%s.weight_quantizer" % l.name
        ref_alpha = g["final/alpha/" + key]
        got = l.weight_quantizer.alpha.detach().numpy()
        dis = np.nonzero((got >= 0) != (ref_alpha >= 0))
        for i in zip(*dis):
            bad.append((l.name, i, float(ref_alpha[i]), float(got[i])))
        wq = l.weight_quantizer
        with torch.no_grad():
            codes = torch.clamp(torch.floor(l.weight / wq.delta) + (wq.alpha >= 0).float() + wq.zero_point, 0, wq.n_levels - 1)
        assert (codes.numpy().astype(np.int16) != g["final/codes/" + key]).sum() == len(dis[0])
    print(fixture, "hard-rounding disagreements:", bad)
    # 100 % agreement except weights whose alpha ENDS within a tenth of one Adam step (lr_w = 5e-2) of zero in both runs: the
    # oracle's closed-form gradients and autograd differ in the last bits, Adam normalises, and twelve steps of +-lr put a
    # handful of the 27 k alphas that close to the rounding boundary
    assert all(abs(r) < 5e-3 and abs(o) < 5e-3 for _, _, r, o in bad) and len(bad) <= 8, bad
