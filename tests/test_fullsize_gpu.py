"""-m gpu: parity at BASELINE.json's full size.  The 400.9M-parameter LDM-4 UNet (random-init weights, W4A8 scales from
16 synthetic calibration rows, as bench.py builds it) is run on 32 rows -- enough rows for every K4 kernel structure
(8-wave, persistent, 4-wave), the fused quantised-output epilogues and the in-place skip concatenation to be the ones
that execute -- once as the fake-quant module graph (the calibration-time forward: K1 fake-quant + fp32 contraction,
the path pinned to the reference's golden outputs on the fixture nets) and once as the frozen int8 engine.
Bound: as for the fixture nets, in units of the output range (an activation within ~1e-6 of a rounding boundary
flips one code; 118 quantised layers deep the flips accumulate: max <= 10 %, mean <= 1 % of the range), plus
two size-independent properties of the engine: the same rows in a different batch position give the same
output bits (row independence: no cross-row term, deterministic kernels), and replays are bit-identical."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def test_ldm4_engine_matches_fake_quant_graph_at_full_size():
    import bench
    dev = torch.device("cuda", 0)
    qnn, _, _ = bench.build_quantised_unet(dev, calib_rows=16)
    g = torch.Generator().manual_seed(77)
    B = 32
    x = torch.randn(B, 3, 64, 64, generator=g).to(dev)
    t = torch.tensor(np.random.RandomState(5).choice(np.arange(0, 1000, 50) + 1, B), dtype=torch.long, device=dev)
    c = torch.randn(B, 1, 512, generator=g).to(dev)
    with torch.no_grad():
        qnn.set_quant_state(True, True)
        fq = qnn(x, t, c).float()
        eng = qnn.freeze()
        out = qnn(x, t, c).float()
        assert qnn.engine is not None
        rng = float(fq.abs().max())
        err = (out - fq).abs()
        print("full-size LDM-4, %d rows: max err %.3e of range, mean %.3e of range" % (B, float(err.max()) / rng, float(err.mean()) / rng))
        assert float(err.max()) <= 0.10 * rng and float(err.mean()) <= 0.01 * rng
        # replay: bit-identical
        assert torch.equal(qnn(x, t, c).float(), out)
        # row independence: rows 0..15 alone (different tile positions, different kernel choices for the smaller M)
        perm = torch.arange(B - 1, -1, -1, device=dev)
        out_p = qnn(x[perm], t[perm], c[perm]).float()
        d = (out_p[perm] - out).abs()
        assert float(d.max()) == 0.0, float(d.max())


def test_ddim_loop_with_all_hoists_is_bit_identical_to_plain_stepping_at_full_size():
    """The compiled sampling loop (HIP-graph replay; cross-attention vectors once per batch, time-embedding table once per
    run, attention-free prefix and first self-attention once per guidance pair, fused producers) against the plain
    loop that calls the eager engine on the doubled batch every step: the same latent bits after 4 DDIM steps on the
    full-size LDM-4 (8 images, CFG)."""
    import bench
    from edadm import ops
    from edadm.sampling import DDIMLoop
    dev = torch.device("cuda", 0)
    qnn, _, _ = bench.build_quantised_unet(dev, calib_rows=16)
    B = 8
    g = torch.Generator().manual_seed(3)
    x_T = torch.randn(B, 3, 64, 64, generator=g).to(dev)
    cond = torch.randn(B, 1, 512, generator=g).to(dev)
    uncond = torch.randn(1, 1, 512, generator=g).expand(B, 1, 512).contiguous().to(dev)
    with torch.no_grad():
        eng = qnn.freeze()
        loop = DDIMLoop(eng, (3, 64, 64), B, steps=4, eta=0.0, scale=3.0, context_shape=(1, 512), device=dev)
        assert loop.unet.ctx_graph is not None and loop.unet.emb_graph is not None
        assert eng.pair_stats["prefix_blocks"] >= 3 and eng.pair_stats["half_attention_blocks"] == 1
        fast = loop.sample(x_T, cond, uncond)
        fast2 = loop.sample(x_T, cond, uncond)
        # plain stepping: eager engine, doubled batch, same update kernel
        img = x_T
        ctx = torch.cat([uncond, cond])
        total = loop.ddim_timesteps.shape[0]
        for i, step in enumerate(np.flip(loop.ddim_timesteps)):
            index = total - i - 1
            ts = torch.full((2 * B,), int(step), device=dev, dtype=torch.long)
            e = eng(torch.cat([img, img]), ts, ctx)
            coef = loop.coef[index:index + 1].expand(B, 5).contiguous()
            img = ops.ddim_step(img.contiguous(), e[B:], e[:B], loop.scale, coef)
    assert torch.isfinite(fast).all()
    assert torch.equal(fast, fast2)
    assert torch.equal(fast, img), float((fast - img).abs().max())
