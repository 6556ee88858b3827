"""CPU: host-side logic of the API mirror — module rewriting, unit traversal order, bit-width
assignment, state-dict compatibility — against structure captured from the reference (G11 inside
the g13_* fixtures).  No compute kernels are called (the product has no CPU compute path)."""
import sys

import numpy as np
import pytest
import torch

from helpers import build_cifar, build_ldm, WQ4, AQ8


def _wrap(g, kind):
    from qdiff import QuantModel
    model = build_cifar(g) if kind == "cifar" else build_ldm(g)      # strict load_state_dict inside
    qnn = QuantModel(model, WQ4, AQ8, sm_abit=8)
    qnn.eval()
    qnn.set_quant_state(False, False)
    qnn.set_first_last_layer_to_8bit()
    qnn.disable_network_output_quantization()
    return qnn


@pytest.mark.parametrize("kind,fixture", [("cifar", "g13_cifar_unet"), ("imagenet", "g13_ldm_imagenet"),
                                           ("church", "g13_ldm_church")])
def test_unit_order_and_bitwidths(golden, kind, fixture):
    from qdiff.quant_layer import QuantModule, UniformAffineQuantizer
    import qdiff  # noqa: F401
    rb = sys.modules['qdiff.recon_block_Qmodel']
    g = golden(fixture)
    qnn = _wrap(g, kind)
    rec = []
    ob, ol = rb.block_reconstruction, rb.layer_reconstruction
    rb.block_reconstruction = lambda m, blk, **k: rec.append(("block", blk))
    rb.layer_reconstruction = lambda m, lay, **k: rec.append(("layer", lay))
    try:
        rb.recon_block_Qmodel(None, qnn, None, {}).recon()
    finally:
        rb.block_reconstruction, rb.layer_reconstruction = ob, ol
    names = {m: n for n, m in qnn.named_modules()}
    got = ["%s:%s:%s" % (k, names[m], type(m).__name__) for k, m in rec]
    assert got == [str(u) for u in g["units"]]
    # bit widths: first / last weight quantizer and the second-to-last act quantizer are 8 bit
    n = 0
    for name, m in qnn.named_modules():
        if isinstance(m, UniformAffineQuantizer):
            k = "qp/%s/n_bits" % name
            if k in g.files:
                assert m.n_bits == int(g[k]), name
                n += 1
    assert n > 50
    mods = [m for m in qnn.model.modules() if isinstance(m, QuantModule)]
    assert mods[-1].disable_act_quant and not any(m.disable_act_quant for m in mods[:-1])
    # every quantizer the reference initialised exists under the same module path (split ones appear
    # after the first split forward, which needs the device)
    mine = {n for n, m in qnn.named_modules() if isinstance(m, UniformAffineQuantizer)}
    ref = {k[3:-len("/delta")] for k in g.files if k.startswith("qp/") and k.endswith("/delta")}
    missing = {r for r in ref if r not in mine and not r.endswith("_0")}
    assert not missing, sorted(missing)[:5]


def test_quant_state_flags_and_engine_dispatch(golden):
    from qdiff.quant_layer import QuantModule
    from qdiff.quant_block import BaseQuantBlock
    qnn = _wrap(golden("g13_ldm_imagenet"), "imagenet")
    qnn.set_quant_state(True, False)
    assert all(m.use_weight_quant and not m.use_act_quant for m in qnn.modules() if isinstance(m, QuantModule))
    qnn.set_quant_state(True, True)
    assert all(m.use_act_quant for m in qnn.modules() if isinstance(m, (QuantModule, BaseQuantBlock)))
    assert qnn.engine is None and qnn.block_count == 0
    qnn.set_grad_ckpt(False)


def test_split_bookkeeping():
    import torch.nn as nn
    from qdiff.quant_layer import QuantModule
    qm = QuantModule(nn.Conv2d(64, 32, 1), WQ4, AQ8)
    assert qm.split == 0 and not hasattr(qm, "act_quantizer_0")
    qm.split = 32
    qm.set_split()
    assert qm.act_quantizer_0.n_bits == 8 and qm.weight_quantizer_0.channel_wise
    with pytest.raises(AssertionError):
        qm.forward(torch.zeros(1, 64, 2, 2), split=16)      # inconsistent split (quant_layer.py:408)


def test_adaround_modes_and_loss_api():
    from edadm.recon import LinearTempDecay, LossFunction
    td = LinearTempDecay(100, rel_start_decay=0.2, start_b=20, end_b=2)
    assert td(10) == 20 and td(100) == 2 and abs(td(60) - 11.0) < 1e-9
    lf = LossFunction(None, round_loss='none', rec_loss='bogus')
    with pytest.raises(ValueError):
        lf(torch.zeros(2, 2), torch.zeros(2, 2))


def test_schedule_helpers(golden):
    from edadm.schedule import make_beta_schedule, make_ddim_timesteps, make_ddim_sampling_parameters, ddim_coef_table
    g = golden("g10_steps")
    b = make_beta_schedule("linear", 1000, linear_start=0.0015, linear_end=0.0195)
    np.testing.assert_array_equal(b, g["ldm/betas"])
    ac = np.cumprod(1.0 - b, axis=0)
    for S in (20, 50):
        ts = make_ddim_timesteps("uniform", S, 1000, verbose=False)
        np.testing.assert_array_equal(ts, g["ldm/S%d/ts" % S])
        sig, al, alp = make_ddim_sampling_parameters(ac, ts, 0.0, verbose=False)
        np.testing.assert_array_equal(al, g["ldm/S%d/alphas" % S])
        np.testing.assert_array_equal(alp, g["ldm/S%d/alphas_prev" % S])
        assert ddim_coef_table(al, alp, sig).shape == (S, 5)


def test_tdac_allocation_matches_reference(golden):
    """G9: density / variety scores and the integer allocation with both fix-up variants."""
    from edadm.tdac import tdac_allocate
    g = golden("g9_tdac")
    for key in sorted({k.split("/")[0] for k in g.files}):
        fm = list(torch.as_tensor(g[key + "/fm"]))
        for variant in ("gt", "ge"):
            dense, cd, w, t_num = tdac_allocate(fm, float(g[key + "/lam"]), int(g[key + "/N"]), float(g[key + "/r"]),
                                                fixup_ge=(variant == "ge"))
            np.testing.assert_array_equal(dense.numpy(), g[key + "/dense_num"])
            np.testing.assert_allclose(cd.numpy(), g[key + "/cos_dis"], rtol=1e-5)
            np.testing.assert_array_equal(t_num.numpy(), g[key + "/t_num_" + variant])


def test_layer_recon_walk_order(golden):
    """--layer_recon: recon_layer_Qmodel takes the blocks apart exactly as the reference does
    (recon_layer_Qmodel.py:20-120; `layer_units` captured by running the reference's walk with stubbed unit functions)."""
    import qdiff  # noqa: F401
    rl = sys.modules['qdiff.recon_layer_Qmodel']
    g = golden("g13_cifar_unet")
    qnn = _wrap(g, "cifar")
    rec = []
    ol, oa = rl.layer_reconstruction, rl.AttnBlock_layer_reconstruction
    rl.layer_reconstruction = lambda m, lay, **k: rec.append(("layer", lay))
    rl.AttnBlock_layer_reconstruction = lambda m, blk, **k: rec.append(("attn", blk))
    try:
        rl.recon_layer_Qmodel(None, qnn, None, {}).recon()
    finally:
        rl.layer_reconstruction, rl.AttnBlock_layer_reconstruction = ol, oa
    names = {m: n for n, m in qnn.named_modules()}
    assert ["%s:%s" % (k, names[m]) for k, m in rec] == [str(u) for u in g["layer_units"]]


def test_sd_shaped_unit_order_conditional_walk(golden):
    """BASELINE config 5: the conditional walk (qdiff_control/recon_block_Qmodel.py:10-43) over the SD-shaped UNet."""
    from helpers import build_ldm_formula
    from qdiff import QuantModel
    import qdiff_control  # noqa: F401
    rb = sys.modules['qdiff_control.recon_block_Qmodel']
    g = golden("g13_ldm_sd")
    qnn = QuantModel(build_ldm_formula(g), WQ4, AQ8, sm_abit=8)
    rec = []
    ob, ol = rb.block_reconstruction, rb.layer_reconstruction
    rb.block_reconstruction = lambda m, blk, **k: rec.append(("block", blk))
    rb.layer_reconstruction = lambda m, lay, **k: rec.append(("layer", lay))
    try:
        rb.recon_block_Qmodel(None, qnn, None, {}).recon()
    finally:
        rb.block_reconstruction, rb.layer_reconstruction = ob, ol
    names = {m: n for n, m in qnn.named_modules()}
    assert ["%s:%s:%s" % (k, names[m], type(m).__name__) for k, m in rec] == [str(u) for u in g["units"]]


def test_attention_step_sizes_trained_by_the_right_walk(golden):
    """Which attention step sizes join the trainables: qdiff/block_recon.py:66-93 knows QuantAttentionBlock and
    QuantAttnBlock, qdiff_control/block_recon.py:68-110 QuantAttnBlock and QuantBasicTransformerBlock."""
    from edadm.recon import _attention_quantizers
    from qdiff.quant_block import QuantBasicTransformerBlock, QuantAttnBlock
    qnn = _wrap(golden("g13_ldm_imagenet"), "imagenet")
    tb = [m for m in qnn.modules() if isinstance(m, QuantBasicTransformerBlock)]
    assert tb
    assert len(_attention_quantizers(tb[0], control=True)) == 8 and _attention_quantizers(tb[0], control=False) == []
    qc = _wrap(golden("g13_cifar_unet"), "cifar")
    ab = [m for m in qc.modules() if isinstance(m, QuantAttnBlock)]
    assert len(_attention_quantizers(ab[0], control=True)) == 4 and len(_attention_quantizers(ab[0], control=False)) == 4


def test_new_attention_order_is_refused_by_the_executor_and_uses_library_operators():
    """openaimodel.py:413-444 (`use_new_attention_order`): not shipped, not quantised by the reference's hooks.  The module's products go
    through the library's operators (edadm.train_ops), never torch.einsum / torch.softmax, and the frozen engine refuses the network."""
    import inspect
    from edadm.nets import ldm_unet
    src = inspect.getsource(ldm_unet.QKVAttention)
    assert "torch.einsum" not in src and "torch.softmax" not in src and "T.bmm_nt" in src and "T.softmax" in src
    from edadm import engine
    eng_src = inspect.getsource(engine)
    assert 'type(m).__name__ == "QKVAttention"' in eng_src and "NotImplementedError" in eng_src
