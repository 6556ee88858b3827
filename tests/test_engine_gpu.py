"""-m gpu: the quantised UNet forward (H2).  Two product paths are checked against the reference's
own outputs (golden g13_*): the fake-quant module graph (calibration-time forward, HIP K1/K2 +
torch contraction) and the frozen int8 executor (all-HIP).  Tolerance: the reference accumulates
fp32 products of de-quantised operands, the executor accumulates exact integers and scales once;
per-layer that is ~1e-6 relative, but an activation that lands within that distance of a rounding
boundary flips one integer code and moves downstream values by one quantisation step, so the
whole-network bound is stated in units of the output's own range."""
import numpy as np
import pytest
import torch

from helpers import build_cifar, build_ldm, quantize_like_reference

pytestmark = pytest.mark.gpu


def _cmp(name, got, ref, tol_max, tol_mean):
    got = got.detach().cpu().numpy().astype(np.float64)
    rng = np.abs(ref).max()
    err = np.abs(got - ref)
    print("%s: max err %.3e (%.2e of range), mean err %.3e" % (name, err.max(), err.max() / rng, err.mean()))
    assert err.max() <= tol_max * rng and err.mean() <= tol_mean * rng, name


@pytest.mark.parametrize("kind", ["cifar", "imagenet", "church"])
def test_quantised_forward_matches_reference(golden, kind):
    g = golden("g13_cifar_unet" if kind == "cifar" else "g13_ldm_%s" % kind)
    model = build_cifar(g) if kind == "cifar" else build_ldm(g)
    qnn, (x, t, ctx), n = quantize_like_reference(model, g, "cifar" if kind == "cifar" else "ldm")
    assert n == len([k for k in g.files if k.startswith("qp/") and k.endswith("/delta")])
    with torch.no_grad():
        _cmp("fp graph", qnn(x, t, ctx), g["out_fp"], 1e-4, 1e-5)
        qnn.set_quant_state(True, False)
        _cmp("weight-quant graph", qnn(x, t, ctx), g["out_wq"], 2e-3, 2e-4)
        qnn.set_quant_state(True, True)
        fq = qnn(x, t, ctx)
        _cmp("fake-quant graph", fq, g["out_q"], 5e-2, 5e-3)
        eng = qnn.freeze()
        modes = {}
        for L in eng.layers.values():
            modes[L.mode] = modes.get(L.mode, 0) + 1
        print("engine layer modes:", modes)
        out = qnn(x, t, ctx)
        assert qnn.engine is not None
        _cmp("int8 engine vs reference", out, g["out_q"], 5e-2, 5e-3)
        _cmp("int8 engine vs fake-quant graph", out, fq.cpu().numpy(), 5e-2, 5e-3)


def test_one_token_context_shortcut_is_bit_exact(golden):
    """Class-conditional LDM: the context is ONE token, softmax over a single key is exactly 1 for every query, so the
    cross-attention branch is one vector per image.  The engine computes it for one query row per image and
    broadcasts (edadm_add_rowbcast); evaluating it for all tokens must give the same bits."""
    g = golden("g13_ldm_imagenet")
    qnn, (x, t, ctx), _ = quantize_like_reference(build_ldm(g), g, "ldm")
    assert ctx.shape[1] == 1
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        eng = qnn.freeze()
        assert eng.one_token_context
        short = eng(x, t, ctx)
        eng.one_token_context = False
        full = eng(x, t, ctx)
    assert torch.equal(short, full)
    with torch.no_grad():                      # ... and with the broadcast add as its own pass instead of inside norm3
        eng.one_token_context, eng.fuse_rowadd_ln = True, False
        assert torch.equal(eng(x, t, ctx), short)


def test_context_branches_hoisted_out_of_the_step_graph_bit_exact(golden):
    """sampling.GraphedUNet computes the one-token cross-attention vectors in a graph of their own, replayed when the
    context changes, and the per-step graph only adds them: same bits as the eager engine evaluating the branch inside
    every call (the softmax over one key is exactly 1 whatever the query), for successive contexts and timesteps."""
    from edadm.sampling import GraphedUNet
    g = golden("g13_ldm_imagenet")
    qnn, (x, t, c), _ = quantize_like_reference(build_ldm(g), g, "ldm")
    assert c.shape[1] == 1
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        eng = qnn.freeze()
        assert eng.context_branches(c) is not None and eng.ctx_r is None
        gu = GraphedUNet(eng, x, t, c)
        assert gu.ctx_graph is not None and eng.ctx_r is None
        for k in range(3):
            ck = (c * (1.0 + 0.5 * k)).contiguous()
            for tt in (t, (t + 37) % 1000):
                ref = eng(x, tt, ck)
                got = gu(x, tt, ck)
                assert torch.equal(got, ref), (k, float((got - ref).abs().max()))
        # an in-place update of the same context tensor is noticed too
        ck.mul_(0.5)
        assert torch.equal(gu(x, t, ck), eng(x, t, ck))
        assert eng.context_branches(torch.cat([c, c], 1)) is None                   # more than one token: no shortcut


def test_time_embedding_table_of_a_schedule_bit_exact(golden):
    """sampling.GraphedUNet with a schedule: the time-embedding path of all steps runs once per sampling run
    (Engine.emb_tables, steps * B rows through the same kernels) and each step copies its row: the same bits as the
    eager engine computing the projections inside the call, for every step of the schedule."""
    from edadm.sampling import GraphedUNet
    g = golden("g13_ldm_imagenet")
    qnn, (x, t, c), _ = quantize_like_reference(build_ldm(g), g, "ldm")
    qnn.set_quant_state(True, True)
    sched = [901, 651, 401, 151, 1]
    with torch.no_grad():
        eng = qnn.freeze()
        gu = GraphedUNet(eng, x, t, c, timesteps=sched)
        assert gu.emb_graph is not None and eng.emb_r is None and eng.ctx_r is None
        for rep in range(2):
            gu.begin()
            for i, ts in enumerate(sched):
                tt = torch.full_like(t, ts)
                assert torch.equal(gu(x, tt, c, step=i), eng(x, tt, c)), (rep, i)
        with pytest.raises(ValueError):
            gu(x, t, c)


def test_skip_conv_operand_from_the_groupnorm_pass_bit_exact(golden):
    """A ResBlock's skip convolution quantises the block input, which the first GroupNorm reads anyway: the GroupNorm apply
    pass writes that operand too (edadm_groupnorm_apply_cat_raw, split quantisers over the skip concatenation included)
    -- the same bits as the separate edadm_quant_i8_cat pass."""
    for name, kind in (("g13_ldm_imagenet", "ldm"),):
        g = golden(name)
        qnn, (x, t, c), _ = quantize_like_reference(build_ldm(g), g, kind)
        qnn.set_quant_state(True, True)
        with torch.no_grad():
            eng = qnn.freeze()
            assert eng.fuse_skip_quant
            fused = eng(x, t, c)
            eng.fuse_skip_quant = False
            plain = eng(x, t, c)
        assert torch.equal(fused, plain)


def test_cfg_pair_shares_the_context_independent_prefix_bit_exact(golden):
    """Classifier-free guidance evaluates [x, x] with contexts [uncond, cond]; the leading blocks without attention do
    not see the context, so sampling.GraphedUNet(cfg_pair=True) runs them once for both halves and keeps their skip
    tensors at half the batch (read periodically by GroupNorm / quantise over the skip concatenation): the same bits
    as the eager engine on the doubled batch."""
    from edadm.sampling import GraphedUNet
    g = golden("g13_ldm_imagenet")
    qnn, (x, t, c), _ = quantize_like_reference(build_ldm(g), g, "ldm")
    qnn.set_quant_state(True, True)
    half = x.shape[0] // 2
    with torch.no_grad():
        eng = qnn.freeze()
        x2 = torch.cat([x[:half], x[:half]]).contiguous()
        t2 = torch.cat([t[:half], t[:half]]).contiguous()
        gu = GraphedUNet(eng, x2, t2, c, cfg_pair=True)
        assert eng.cfg_pair is False
        # the graph shares the attention-free prefix AND the first attention block up to its self-attention
        assert eng.pair_stats["prefix_blocks"] >= 1 and eng.pair_stats["half_attention_blocks"] == 1
        for k in range(2):
            xk = torch.cat([x[k:k + half], x[k:k + half]]).contiguous() if k + half <= x.shape[0] else x2
            ck = (c * (1.0 + 0.3 * k)).contiguous()
            assert torch.equal(gu(xk, t2, ck), eng(xk, t2, ck)), k
        for flag in (True, False):                       # with and without the fused skip-convolution operand
            eng.fuse_skip_quant = flag
            eng.cfg_pair = True
            try:
                shared = eng(x2, t2, c)
            finally:
                eng.cfg_pair = False
            assert torch.equal(shared, eng(x2, t2, c)), flag
