"""-m gpu: the calibration hot loop (H1) through the product API on the device, against the
reference's own trajectories (golden g8_recon: prob = input_prob = 1, python `random` fixes idx):
scale initialisation (K3), activation caching, and 12 iterations of layer / block reconstruction
per unit with alpha and delta recorded after every Adam step."""
import random

import numpy as np
import pytest
import torch

from helpers import build_toynet, WQ4, AQ8

pytestmark = pytest.mark.gpu


def test_scale_init_cache_and_reconstruction(golden):
    from qdiff import QuantModel, set_weight_quantize_params, set_act_quantize_params
    from qdiff.data_utils import save_inp_oup_data
    from qdiff.block_recon import block_reconstruction
    from qdiff.layer_recon import layer_reconstruction
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.quant_layer import UniformAffineQuantizer
    import edadm.recon as recon
    g = golden("g8_recon")
    aq = dict(AQ8)
    aq["prob"] = 1.0
    qnn = QuantModel(build_toynet(g), WQ4, aq, sm_abit=8).cuda().eval()
    x, t = torch.as_tensor(g["x"]).cuda(), torch.as_tensor(g["t"]).cuda()
    cali = (x, t)
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali, batch_size=32)
    # ---- K3: every delta / zero point the reference derived, from the HIP search
    n = 0
    for name, m in qnn.named_modules():
        if isinstance(m, UniformAffineQuantizer) and m.delta is not None:
            k = "init/qp/" + name
            ref_d, ref_z = g[k + "/delta"].reshape(-1), g[k + "/zero_point"].reshape(-1)
            got_d, got_z = m.delta.detach().cpu().numpy().reshape(-1), m.zero_point.cpu().numpy().reshape(-1)
            if m.leaf_param:
                # activations reach the quantizer through GPU convolutions (fp32 sums in another order than
                # the CPU reference): a near-tie between two of the 100 candidates (1 % apart) can flip in one
                # of the two EMA batches (weight 0.1) -> one or two grid steps (1 % each) at worst; bit-exact on identical
                # inputs is shown by tests/test_quantizer_gpu.py
                np.testing.assert_allclose(got_d, ref_d, rtol=2.5e-2)
                assert np.abs(got_z - ref_z).max() <= 1
            else:                # weights: bit-exact scales and zero points
                np.testing.assert_array_equal(got_d, ref_d)
                np.testing.assert_array_equal(got_z, ref_z)
            n += 1
    assert n == len([k for k in g.files if k.startswith("init/qp/") and k.endswith("/delta")])
    # ---- a9: cached activations
    qnn.set_quant_state(True, True)
    res, ci, co = save_inp_oup_data(qnn, qnn.model.rb, cali, True, True, batch_size=32, input_prob=True, keep_gpu=False)
    assert res == bool(g["g12/rb/resblock"])
    np.testing.assert_allclose(ci[1][0].cpu().numpy(), g["g12/rb/inp_fp"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ci[1][1].cpu().numpy(), g["g12/rb/temb_fp"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(co.cpu().numpy(), g["g12/rb/out_fp"], rtol=1e-4, atol=1e-5)
    dq = np.abs(ci[0][0].cpu().numpy() - g["g12/rb/inp_q"])
    assert np.median(dq) < 1e-5 and dq.max() < 0.2          # quantised prefix: identical up to rare code flips
    # ---- oracle twin started from the PRODUCT's own initial scales (separates init-grid chaos from loop bugs)
    from oracle import qdiff_oracle as O
    from test_oracle_nets import ToyNet as OToyNet, sub_sd as o_sub_sd
    onet = OToyNet(o_sub_sd(g, "sd/"), WQ4, aq)
    with torch.no_grad():
        onet(x.cpu(), t.cpu())
    st = {}
    for name, m in qnn.named_modules():
        if isinstance(m, UniformAffineQuantizer) and m.delta is not None:
            st["qp/" + name + "/delta"] = m.delta.detach().cpu().numpy()
            st["qp/" + name + "/zero_point"] = m.zero_point.cpu().numpy()
            st["qp/" + name + "/n_bits"] = np.int64(m.n_bits)
    onet.load_qparams(st)
    # ---- a6 / a7: trajectories
    kwargs = dict(cali_data=cali, iters=12, act_quant=True, asym=True, opt_mode="mse", lr_a=1e-3, lr_w=5e-2, p=2.0,
                  weight=0.0001, b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=1.0, add_loss=0.8,
                  recon_w=True, recon_a=True, keep_gpu=True)
    traj, cur = {}, {"name": None}
    orig = recon.FusedAdam.launch

    def step(self):
        orig(self)
        key = "%s/%s" % (cur["name"], "a" if self.params[0].numel() == 1 else "w")
        traj.setdefault(key, []).append(self.flat.detach().cpu().clone())

    recon.FusedAdam.launch = step
    try:
        random.seed(8080)
        for name, fn in (("conv_in", layer_reconstruction), ("temb_lin", layer_reconstruction),
                         ("rb", block_reconstruction), ("at", block_reconstruction), ("conv_out", layer_reconstruction)):
            cur["name"] = name
            fn(qnn, getattr(qnn.model, name), **kwargs)
    finally:
        recon.FusedAdam.launch = orig
    # the same five reconstructions in the oracle (CPU), same idx stream
    otraj = {}
    random.seed(8080)
    okw = dict(cali=(x.cpu(), t.cpu()), iters=12, act_quant=True, lr_a=1e-3, lr_w=5e-2, p=2.0, batch_size=16,
               input_prob=1.0, add_loss=0.8, recon_w=True, recon_a=True, cache_batch=32)
    for name, kind in (("conv_in", "layer"), ("temb_lin", "layer"), ("rb", "block"), ("at", "block"), ("conv_out", "layer")):
        tw, ta = [], []
        O.reconstruct_unit(onet, getattr(onet, name), kind,
                           trace=lambda it, wp, ap, l: (tw.append(torch.cat([p.detach().flatten() for p in wp]).clone()),
                                                        ta.append(torch.cat([p.detach().flatten() for p in ap]).clone())),
                           **okw)
        otraj[name] = (torch.stack(tw).numpy(), torch.stack(ta).numpy())
    stats = []
    for name in ("conv_in", "temb_lin", "rb", "at", "conv_out"):
        ref_w, ref_a = otraj[name]
        gold_w = g["g8/traj/%s/w" % name]
        got_w, got_a = torch.stack(traj[name + "/w"]).numpy(), torch.stack(traj[name + "/a"]).numpy()
        dw = np.abs(got_w - ref_w)
        print(name, "vs reference golden: median %.2e" % np.median(np.abs(got_w - gold_w)))
        print(name, "alpha: median %.2e frac>1e-2 %.4f max %.3g | delta max rel %.3g" % (
            np.median(dw), (dw > 1e-2).mean(), dw.max(), (np.abs(got_a - ref_a) / np.abs(ref_a)).max()))
        # lr_w = 5e-2 per step: the trajectories coincide to <1 % of a step; an element whose gradient is at
        # rounding-noise level may take a different +-lr step (Adam normalises by sqrt(v)): <2 % tail
        stats.append((name, np.median(dw), (dw > 1e-2).mean(), dw.max(), np.median(dw[0]), (np.abs(got_a - ref_a) / np.abs(ref_a)).max()))
    # lr_w = 5e-2 per Adam step.  The first step of every unit is identical (same gradient signs); later
    # steps drift because the activation step sizes start a grid step away (see above) and Adam normalises
    # near-zero gradients to +-lr: the trajectories stay within a few % of one step in the median
    for name, med, frac, mx, med0, arel in stats:
        assert med0 < 1e-5, (name, med0)
        assert med < 4e-3 and frac < 0.2 and mx < 0.25, (name, med, frac, mx)   # the 5th unit carries every upstream code flip
        assert arel < 3e-2, (name, arel)
    assert qnn.block_count == int(g["g8/block_count"])
    # final hard rounding decisions: the integer weights agree
    for name, m in qnn.named_modules():
        if isinstance(m, AdaRoundQuantizer):
            ref = g["g8/final/alpha/" + name]
            agree = np.mean((m.alpha.detach().cpu().numpy() >= 0) == (ref >= 0))
            print("G8 final hard rounding", name, "agree %.5f (%d of %d differ)" % (agree, int(round((1 - agree) * ref.size)), ref.size))
            # measured (round 4): 10 of 9216 at worst (rb.conv1); gate at 2x
            assert int(round((1 - agree) * ref.size)) <= max(2, int(0.0022 * ref.size)), (name, agree)
            assert m.soft_targets is False
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        out = qnn(x[:8], t[:8]).cpu().numpy()
    ref = g["g8/final/out_q"]
    assert np.abs(out - ref).max() < 0.08 * np.abs(ref).max()


def test_fp_feature_cache_equals_per_iteration_fp_forward(golden):
    """edadm/recon.py fp_features: the per-sample FP feature maps computed once equal what the per-iteration FP forward
    of block_recon.py:170-178 yields for any drawn batch (rows are independent of the batch they ride in; the only
    batch-dependent quantity is the f16 expansion's power-of-two scale, far below the tolerance)."""
    from qdiff import QuantModel, set_weight_quantize_params, set_act_quantize_params
    from qdiff.data_utils import save_inp_oup_data
    from qdiff.quant_layer import QuantModule
    from qdiff.utils import AttentionMap
    from edadm import recon
    g = golden("g8_recon")
    aq = dict(AQ8)
    aq["prob"] = 1.0
    qnn = QuantModel(build_toynet(g), WQ4, aq, sm_abit=8).cuda().eval()
    x, t = torch.as_tensor(g["x"]).cuda(), torch.as_tensor(g["t"]).cuda()
    cali = (x, t)
    set_weight_quantize_params(qnn, cali)
    set_act_quantize_params(qnn, cali, batch_size=32)
    for name in ("rb", "at"):
        unit = getattr(qnn.model, name)
        resblock, ci, co = save_inp_oup_data(qnn, unit, cali, True, True, batch_size=32, input_prob=True)
        hooks = [AttentionMap(m) for m in unit.modules() if isinstance(m, QuantModule)]
        sz = co.size(0)
        feats = recon.fp_features(unit, hooks, ci, resblock, sz, 16, 1 << 30)
        assert feats is not None and len(feats) == len(hooks) - 1 and all(f.shape[0] == sz for f in feats)
        assert recon.fp_features(unit, hooks, ci, resblock, sz, 16, 1024) is None          # over budget: recompute path
        idx = torch.tensor(random.Random(5).sample(range(sz), 16), device="cuda")
        unit.set_quant_state(False, False)
        with torch.no_grad():
            unit(*((ci[1][0][idx], ci[1][1][idx]) if resblock else (ci[1][idx],)))
        for f, h in zip(feats, hooks[:-1]):
            ref = h.out
            assert torch.allclose(f[idx], ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max())), name
        for h in hooks:
            h.remove()


def test_graph_replayed_iterations_equal_eager(golden):
    """edadm/recon.py: from the third iteration on, a unit's iteration (minibatch gather, three forwards, autograd's backward,
    both Adam launches) is one HIP-graph replay.  Without stochastic masks (prob = input_prob = 1) the replayed loop must
    leave exactly the alphas and step sizes of the eager loop, bit for bit -- same kernels in the same order -- for a block
    with the fine-grained loss, an attention block and a single layer; with masks (prob 0.5) two graph runs from the same
    seeds agree with each other and differ from iteration to iteration (the device-side epoch)."""
    from qdiff import QuantModel, set_weight_quantize_params, set_act_quantize_params
    from qdiff.block_recon import block_reconstruction
    from qdiff.layer_recon import layer_reconstruction
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.quant_layer import UniformAffineQuantizer, seed_mask_rng
    import edadm.recon as recon
    from edadm import ops
    g = golden("g8_recon")
    x, t = torch.as_tensor(g["x"]).cuda(), torch.as_tensor(g["t"]).cuda()

    took_inject = []

    def run(min_iters, prob, inp_prob):
        aq = dict(AQ8)
        aq["prob"] = prob
        qnn = QuantModel(build_toynet(g), WQ4, aq, sm_abit=8).cuda().eval()
        set_weight_quantize_params(qnn, (x, t))
        set_act_quantize_params(qnn, (x, t), batch_size=32)
        kw = dict(cali_data=(x, t), iters=24, act_quant=True, asym=True, opt_mode="mse", lr_a=1e-4, lr_w=5e-2, p=2.0, weight=0.0001,
                  b_range=(20, 2), warmup=0.2, batch_size=16, input_prob=inp_prob, add_loss=0.8, recon_w=True, recon_a=True)
        old = recon.GRAPH_MIN_ITERS
        recon.GRAPH_MIN_ITERS = min_iters
        try:
            random.seed(77)
            seed_mask_rng(77)
            ops.rng_epoch(0)
            layer_reconstruction(qnn, qnn.model.conv_in, **kw)
            block_reconstruction(qnn, qnn.model.rb, **kw)
            took_inject.append(bool(getattr(qnn.model.rb, "recon_inject", False)))
            block_reconstruction(qnn, qnn.model.at, **kw)
        finally:
            recon.GRAPH_MIN_ITERS = old
            ops.rng_epoch(0)
        out = {}
        for name, m in qnn.named_modules():
            if isinstance(m, AdaRoundQuantizer):
                out[name + "/alpha"] = m.alpha.detach().cpu().clone()
            if isinstance(m, UniformAffineQuantizer) and m.delta is not None and m.leaf_param:
                out[name + "/delta"] = m.delta.detach().cpu().clone()
        return out

    eager, graphed = run(10 ** 9, 1.0, 1.0), run(4, 1.0, 1.0)
    assert eager.keys() == graphed.keys() and len(eager) > 10
    for k in eager:
        assert torch.equal(eager[k], graphed[k]), (k, float((eager[k] - graphed[k]).abs().max()))
    a, b = run(4, 0.5, 0.5), run(4, 0.5, 0.5)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert any(not torch.equal(a[k], graphed[k]) for k in a)             # the masks do something
    # the per-module loss terms as gradient injections (recon.INJECT_MODULE_LOSS, csrc/elem.hip k_lp_inject: one pass instead of
    # gather + loss backward + zero-padded slice gradient + accumulation add) leave the bits of the plain autograd form, with and
    # without masks, eager and replayed
    assert recon.INJECT_MODULE_LOSS
    recon.INJECT_MODULE_LOSS = False
    try:
        plain_eager, plain_masks = run(10 ** 9, 1.0, 1.0), run(4, 0.5, 0.5)
    finally:
        recon.INJECT_MODULE_LOSS = True
    for k in eager:
        assert torch.equal(eager[k], plain_eager[k]), (k, float((eager[k] - plain_eager[k]).abs().max()))
        assert torch.equal(a[k], plain_masks[k]), (k, float((a[k] - plain_masks[k]).abs().max()))
    assert took_inject[:4] == [True] * 4 and took_inject[4:] == [False, False], took_inject   # the residual block took the fused form
