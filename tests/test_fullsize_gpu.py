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


# ---------------------------------------------------------------------------------------------------------------------------
# G20: the reconstruction loop in INDEX SPACE at a size where the production arithmetic runs (tests/golden/_g20.py)
# ---------------------------------------------------------------------------------------------------------------------------
F16X3_ENTRY_POINTS = ("edadm_qgemm_f16x3", "edadm_gemm_f16x3_nt", "edadm_split_f16", "edadm_transpose_split_f16",
                      "edadm_qconv3_f16x3_direct")


def _g20_targets(name, g):
    """FP targets of the cached rows: the unit with quantisation off on the FP inputs (data_utils.py:133-139), evaluated by the
    CPU oracle on this host (plain torch fp32, the reference's own operators) and checked against the samples the reference stored."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _g20
    from _weights import formula_state_dict
    from oracle import qdiff_oracle as O
    from helpers import WQ4, AQ8
    from edadm.nets.ldm_unet import ResBlock, BasicTransformerBlock
    with torch.device("meta"):
        unit = (ResBlock(_g20.RES["channels"], _g20.RES["emb_channels"], 0.0, out_channels=_g20.RES["out_channels"], dims=2)
                if name == "res" else BasicTransformerBlock(_g20.TF["dim"], _g20.TF["heads"], _g20.TF["d_head"],
                                                             context_dim=_g20.TF["context_dim"], gated_ff=True, checkpoint=False))
    sd = formula_state_dict([("%s.%s" % (name, k), tuple(v.shape)) for k, v in unit.state_dict().items()], _g20.SEED)
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    B = O._Builder(sd, WQ4, AQ8, 8)
    ou = (O.OResBlock(B, name, name, _g20.RES["channels"], _g20.RES["out_channels"]) if name == "res"
          else O.OTransformerBlock(B, name, name, _g20.TF["heads"]))
    ou.set_quant_state(False, False)
    cq, cf = _g20.caches(name)
    with torch.no_grad():
        out = torch.cat([ou(torch.from_numpy(cf[0][i:i + 32]), torch.from_numpy(cf[1][i:i + 32])) for i in range(0, _g20.ROWS, 32)])
    pos = torch.from_numpy(_g20.sample_positions(out.numel()))
    ref = torch.from_numpy(g["out_fp/%s/sample" % name])
    got = out.reshape(-1)[pos]
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max()) / scale
    print("G20 %s: FP targets on this host vs the reference's samples: max %.2e of range, %d of %d samples bit-identical"
          % (name, err, int((got == ref).sum()), ref.numel()))
    assert err < 2e-6
    assert abs(float(out.double().sum()) - float(g["out_fp/%s/sum" % name])) <= 1e-6 * float(out.double().abs().sum())
    return cq, cf, out


def _g20_run(name, g, caches, f16x3):
    """one reconstruction of unit `name` with contract.F16X3 = f16x3 on the reference's scales, caches, draws and masks"""
    import random
    import _g20
    import _uniforms
    from _weights import formula_state_dict
    from helpers import WQ4, AQ8
    from qdiff import QuantModel
    from qdiff.adaptive_rounding import AdaRoundQuantizer
    from qdiff.quant_layer import UniformAffineQuantizer
    from edadm.state import load_quant_state
    from edadm.nets.ldm_unet import ResBlock, BasicTransformerBlock
    from edadm import contract, lib
    import edadm.recon as recon
    import torch.nn as nn
    dev = torch.device("cuda", 0)

    class Host(nn.Module):
        def __init__(self):
            super().__init__()
            self.in_channels = _g20.RES["channels"]
            self.res = ResBlock(_g20.RES["channels"], _g20.RES["emb_channels"], 0.0, out_channels=_g20.RES["out_channels"], dims=2,
                                use_checkpoint=False, use_scale_shift_norm=False)
            self.tf = BasicTransformerBlock(_g20.TF["dim"], _g20.TF["heads"], _g20.TF["d_head"], dropout=0.0,
                                            context_dim=_g20.TF["context_dim"], gated_ff=True, checkpoint=False)

    host = Host().eval()
    sd = formula_state_dict([(k, tuple(v.shape)) for k, v in host.state_dict().items()], _g20.SEED)
    host.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    aq = dict(AQ8)
    aq["prob"] = _g20.PROB
    qnn = QuantModel(host, WQ4, aq, sm_abit=8).to(dev).eval()
    qnn.set_grad_ckpt(False)
    pre = "init/qp/"
    keys = {k: g[k] for k in g.files if k.startswith(pre + "model.%s." % name)}
    n = load_quant_state(qnn, keys, prefix=pre)
    assert n == len([k for k in keys if k.endswith("/delta")]), (n, len(keys))
    unit = getattr(qnn.model, name)
    cq, cf, out_fp = caches
    cqd, cfd, ofd = [torch.from_numpy(a).to(dev) for a in cq], [torch.from_numpy(a).to(dev) for a in cf], out_fp.to(dev)

    def save_fn(model, u, cali, asym, act_quant, batch_size=32, input_prob=True, keep_gpu=True):
        return True, ([cqd[0], cqd[1]], [cfd[0], cfd[1]]), ofd

    rep = _uniforms.ReplayHash(device=dev)
    for qn, m in qnn.named_modules():
        if isinstance(m, UniformAffineQuantizer) and m.leaf_param and qn.startswith("model.%s." % name):
            m.injected_uniform = (lambda nm: lambda xx: rep.draw_calls(nm, "iter", xx.shape, recon.STATE["batched"]))(qn)
    recon.INJECT_MIX_UNIFORM = lambda xx: rep.draw("input_mix:" + name, "iter", xx.shape)
    traj_w, traj_a, idx_log = [], [], []
    orig_launch, orig_sample = recon.FusedAdam.launch, random.sample

    grad0 = {}

    def launch(self):
        orig_launch(self)
        key = "a" if self.params[0].numel() == 1 else "w"
        if key not in grad0:                                   # iteration 0: the gradients the first Adam step saw (collected slab)
            grad0[key] = self.grad.detach().clone()
        (traj_a if key == "a" else traj_w).append(self.flat.detach().clone())

    def sample(pop, k):
        r = orig_sample(pop, k)
        idx_log.append(list(r))
        return r

    alpha0 = []
    hyper = {k: v for k, v in _g20.HYPER.items()}
    old = contract.F16X3
    contract.F16X3 = f16x3
    recon.FusedAdam.launch, random.sample = launch, sample
    lib.CALLS = {}
    try:
        random.seed(_g20.SEED + 1)
        recon.reconstruct(qnn, unit, None, is_block=True, iters=int(g["iters"]), control=True, save_fn=save_fn,
                          cache_batch=hyper["batch_size"], **hyper)
        calls = dict(lib.CALLS)
    finally:
        recon.FusedAdam.launch, random.sample = orig_launch, orig_sample
        recon.INJECT_MIX_UNIFORM = None
        contract.F16X3 = old
        lib.CALLS = None
    assert np.array_equal(np.asarray(idx_log), g["idx/" + name])                    # the reference's minibatch draws
    order = [qn for qn, m in unit.named_modules() if isinstance(m, AdaRoundQuantizer)]
    assert order == [str(s) for s in g["order/%s/w" % name]], order
    for qn, m in unit.named_modules():
        if isinstance(m, AdaRoundQuantizer):
            w = dict(unit.named_modules())[qn.rsplit(".", 1)[0]].org_weight
            with torch.no_grad():
                rest = (w / m.delta) - torch.floor(w / m.delta)
                alpha0.append((-torch.log((m.zeta - m.gamma) / (rest - m.gamma) - 1)).flatten())
    got_log = sorted("%s|%s|%d|%s" % (o, p, c, "x".join(map(str, s))) for o, p, c, s in rep.log)
    tw, ta = torch.stack(traj_w), torch.stack(traj_a)
    return dict(tw=tw, ta=ta.cpu().numpy(), alpha0=torch.cat(alpha0), calls=calls, log=got_log, batched=recon.STATE["batched"],
                g0w=grad0["w"].cpu().double().numpy(), g0a=grad0["a"].cpu().double().numpy(),
                sizes=[(qn, m.alpha.numel()) for qn, m in unit.named_modules() if isinstance(m, AdaRoundQuantizer)],
                a_names=[qn for qn, m in unit.named_modules() if isinstance(m, UniformAffineQuantizer) and m.leaf_param])


@pytest.mark.parametrize("name", ["res", "tf"])
def test_recon_unit_f16x3_vs_exact_fp32_vs_reference(golden, name):
    """VERDICT r3 item 1.  One LDM-4-sized unit (ResBlock 192 -> 384 at 32 x 32; transformer block d = 384, 1024 tokens), 32-row
    minibatches, shipped hyper-parameters and 0.5 / 0.5 masks, reconstructed three ways on IDENTICAL scales, caches, draws
    and masks: the reference on CPU (fixture G20, qdiff_control/block_recon.py:13-243), the product with the three-product
    f16 contraction (production: contract.F16X3 = True) and the product on the exact-fp32 MFMA (F16X3 = False).  Compared
    in index space: the direction of every alpha's first Adam step and every alpha's final hard rounding."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _g20
    g = golden("g20_f16x3_units")
    caches = _g20_targets(name, g)
    runs = {mode: _g20_run(name, g, caches, mode) for mode in (True, False)}
    n = int(g["final/%s/count" % name])
    ref_sign = _g20.unpack(g["final/%s/sign" % name], n)
    ref_near = _g20.unpack(g["final/%s/near" % name], n)
    ref_up = _g20.unpack(g["first/%s/up" % name], n)
    ref_moved = _g20.unpack(g["first/%s/moved" % name], n)
    ref_log = sorted(str(l) for l in g["rand/log"] if ("|iter|" in str(l)) and (("model.%s." % name) in str(l) or str(l).startswith("input_mix:" + name)))
    bad = {}
    for mode, r in runs.items():
        label = "f16x3" if mode else "exact fp32"
        took = sorted(k for k in r["calls"] if k in F16X3_ENTRY_POINTS)
        print("G20 %s [%s]: contraction entry points %s; batched forwards %s" % (name, label, {k: r["calls"][k] for k in took}, r["batched"]))
        if mode:
            need = {"edadm_qgemm_f16x3", "edadm_gemm_f16x3_nt", "edadm_transpose_split_f16", "edadm_split_f16"}
            if name == "res":
                need.add("edadm_qconv3_f16x3_direct")
            assert need <= set(took), (need, took)
        else:
            assert not took, took
        assert r["log"] == ref_log                                         # every mask drawn as the reference drew it
        tw = r["tw"]
        assert tw.shape[1] == n
        a1, a0, af = tw[0], r["alpha0"], tw[-1]
        up = (a1 > a0).cpu().numpy()
        moved = (a1 != a0).cpu().numpy()
        first_bad = int(((up != ref_up) & (moved | ref_moved)).sum())
        sign = (af >= 0).cpu().numpy()
        dis = np.nonzero(sign != ref_sign)[0]
        near_got = (af.abs() < _g20.NEAR).cpu().numpy()
        far = [int(i) for i in dis if not (ref_near[i] and near_got[i])]
        bad[mode] = set(int(i) for i in dis)
        ref_w = g["traj/%s/w" % name]
        dw = np.abs(tw[:, ::_g20.STRIDE].cpu().numpy() - ref_w)
        ref_a = g["traj/%s/a" % name]
        da = np.abs(r["ta"] - ref_a) / np.abs(ref_a)
        worst_q = int(np.argmax(da.max(0)))
        print("G20 %s [%s] vs REFERENCE: first Adam step direction differs on %d of %d alphas; final hard rounding differs on %d "
              "(%d not next to zero in both); strided alpha trajectory median %.2e frac>lr/10 %.5f; delta trajectory max rel %.2e (step size #%d, %.4g)"
              % (name, label, first_bad, n, len(dis), len(far), np.median(dw), (dw > 0.05).mean(), da.max(), worst_q, float(ref_a[0, worst_q])))
        r["first_bad"], r["far"], r["dis"] = first_bad, far, dis
        # iteration-0 gradients against the reference's (round-4 review, item 2c): is the distance to the reference a matter of
        # summation order (relative L2 ~ 1e-6, the reference-vs-itself level) or of an operator that computes something else (1e-4+)?
        rw, ra = g["grad0/%s/w" % name].astype(np.float64), g["grad0/%s/a" % name].astype(np.float64)
        pw, pa = r["g0w"][::_g20.STRIDE], r["g0a"]
        r["gw_rel"] = float(np.linalg.norm(pw - rw) / np.linalg.norm(rw))
        r["ga_rel"] = float(np.linalg.norm(pa - ra) / np.linalg.norm(ra))
        r["ga_worst"] = float((np.abs(pa - ra) / np.abs(ra).max()).max())
        r["gw_norm_rel"] = float(abs(np.linalg.norm(r["g0w"]) - float(g["grad0/%s/w_norm" % name])) / float(g["grad0/%s/w_norm" % name]))
        print("G20 %s [%s] iteration-0 gradients vs REFERENCE: d loss / d alpha rel L2 %.2e over %d strided alphas (norm of all %d: rel %.2e); "
              "d loss / d delta rel L2 %.2e, worst element %.2e of the largest"
              % (name, label, r["gw_rel"], pw.size, r["g0w"].size, r["gw_norm_rel"], r["ga_rel"], r["ga_worst"]))
        # per layer: where the distance sits
        pos, off, rows = np.arange(0, r["g0w"].size, _g20.STRIDE), 0, []
        for qn, k in r["sizes"]:
            sel = (pos >= off) & (pos < off + k)
            if sel.any():
                rows.append("%s %.1e (|g| %.1e)" % (qn.replace(".weight_quantizer", ""), np.linalg.norm(pw[sel] - rw[sel]) / max(np.linalg.norm(rw[sel]), 1e-30),
                                                  np.linalg.norm(rw[sel])))
            off += k
        print("   per layer d alpha rel L2:", "; ".join(rows))
        if len(r["a_names"]) == pa.size:
            print("   per quantiser d delta (product / reference):", "; ".join("%s %.4g/%.4g" % (n_.split(".", 2)[-1], a_, b_) for n_, a_, b_ in zip(r["a_names"], pa, ra)))
        print("   not next to zero:", [(i, float(af[i]), bool(ref_sign[i]), bool(ref_near[i])) for i in far[:8]])
    extra = bad[True] - bad[False]
    print("G20 %s: disagreements with the reference -- f16x3 %d, exact fp32 %d, in f16x3 only %d, in exact only %d"
          % (name, len(bad[True]), len(bad[False]), len(extra), len(bad[False] - bad[True])))
    # the floor: the reference against ITSELF with 3 CPU threads instead of 8 (g20_reference_3threads: another partition of torch's
    # fp32 sums, nothing else changed)
    alt = golden("g20_reference_3threads")
    alt_sign, alt_near = _g20.unpack(alt["final/%s/sign" % name], n), _g20.unpack(alt["final/%s/near" % name], n)
    alt_dis = alt_sign != ref_sign
    alt_a = np.abs(alt["traj/%s/a" % name] - g["traj/%s/a" % name]) / np.abs(g["traj/%s/a" % name])
    print("G20 %s: the reference with 3 threads vs the reference with 8: final hard rounding differs on %d (%d not next to zero in both); "
          "delta trajectory max rel %.2e" % (name, int(alt_dis.sum()), int((alt_dis & ~(alt_near & ref_near)).sum()), alt_a.max()))
    aw, rw8 = alt["grad0/%s/w" % name].astype(np.float64), g["grad0/%s/w" % name].astype(np.float64)
    aa, ra8 = alt["grad0/%s/a" % name].astype(np.float64), g["grad0/%s/a" % name].astype(np.float64)
    floor_w, floor_a = float(np.linalg.norm(aw - rw8) / np.linalg.norm(rw8)), float(np.linalg.norm(aa - ra8) / np.linalg.norm(ra8))
    print("G20 %s: iteration-0 gradients, the reference with 3 threads vs 8: d loss / d alpha rel L2 %.2e, d loss / d delta rel L2 %.2e"
          % (name, floor_w, floor_a))
    # How well is the reference's own gradient defined?  Another thread count hardly moves torch's CPU sums (above); the conditioning
    # probes do: the reference's iteration 0 re-run with every cached input multiplied by 1 + 2^-23 (one unit in the last place) and by
    # 1 + 2^-18 (32 units: what two correct fp32 evaluations of a 400..1500-term dot product in different summation orders differ
    # by), same scales, draws and masks (make_golden.py::g20_ulp_floor / g20_ulp32_floor).  The gradient is a function of pred - target
    # through 8-bit fake-quantisers whose codes flip at .5 boundaries (a softmax code is worth 20 % of a probability), so it moves by
    # 5e-5 (ResBlock) / 5e-4 (transformer block) for ONE ulp and 2e-4 / 2e-2 for 32.  The product's distance -- on the three-product
    # f16 contraction and on the exact-fp32 MFMA alike -- sits inside that band on every count: rounding-level differences amplified by
    # the conditioning of the graph, not an operator that computes something else.  Gate: the reference's own 32-ulp response.
    floors = {}
    for tag, fx in (("1 ulp", "g20_reference_ulp"), ("32 ulp", "g20_reference_ulp32")):
        pf = golden(fx)
        pw_, pa_ = pf["grad0/%s/w" % name].astype(np.float64), pf["grad0/%s/a" % name].astype(np.float64)
        floors[tag] = (float(np.linalg.norm(pw_ - rw8) / np.linalg.norm(rw8)), float(np.linalg.norm(pa_ - ra8) / np.linalg.norm(ra8)))
        print("G20 %s: iteration-0 gradients, the reference with its inputs moved by %s: d loss / d alpha rel L2 %.2e, d loss / d delta rel L2 %.2e"
              % (name, tag, floors[tag][0], floors[tag][1]))
    for mode, r in runs.items():
        assert r["gw_rel"] <= floors["32 ulp"][0] and r["ga_rel"] <= floors["32 ulp"][1], (mode, r["gw_rel"], r["ga_rel"], floors)
    # THE EXACT VALUE (round-5 review, item 3).  Neither the reference's fp32 CPU run nor the product is the truth: fixture
    # g20_reference_fp64 is the reference's own graph on the same fp32 weights, caches, scales, draws and masks with every module and
    # cached tensor in float64 (make_golden.py::g20_fp64_truth).  Distances to it, relative L2 over the same strided alphas / all deltas:
    #   reference fp32 (8 threads) -> fp64: ResBlock 4.0e-5 / 7.2e-6, transformer block 3.99e-3 / 3.85e-4 (computed from the fixtures
    #   alone, CPU test tests/test_g20_fp64_host.py) -- i.e. the REFERENCE is as far from the exact gradient as the product is from the
    #   reference.  The product is gated at 2x the reference's own distance to the exact value.
    f64 = golden("g20_reference_fp64")
    tw64, ta64 = f64["grad0/%s/w" % name].astype(np.float64), f64["grad0/%s/a" % name].astype(np.float64)
    rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))
    ref64 = (rel(rw8, tw64), rel(ra8, ta64))
    print("G20 %s: iteration-0 gradients vs the EXACT (float64) value: reference fp32 d alpha %.2e, d delta %.2e" % (name, ref64[0], ref64[1]))
    for mode, r in runs.items():
        r["gw64"], r["ga64"] = rel(r["g0w"][::_g20.STRIDE], tw64), rel(r["g0a"], ta64)
        # sign of the gradient = direction of the first Adam step: how many of the strided alphas go the other way than exact arithmetic says
        nz = tw64 != 0
        r["sgn64"] = int((np.sign(r["g0w"][::_g20.STRIDE])[nz] != np.sign(tw64)[nz]).sum())
        print("G20 %s [%s] vs the EXACT value: d alpha %.2e (%.2fx the reference's distance), d delta %.2e (%.2fx); gradient sign differs on %d of %d "
              "strided alphas (reference fp32: %d)" % (name, "f16x3" if mode else "exact fp32", r["gw64"], r["gw64"] / ref64[0], r["ga64"],
                                                       r["ga64"] / ref64[1], r["sgn64"], int(nz.sum()), int((np.sign(rw8)[nz] != np.sign(tw64)[nz]).sum())))
    # Measured (round 6, MI355X): d alpha -- ResBlock product 6.5e-5 (1.6x the reference's 4.0e-5), transformer block product 6.6e-4 / 7.2e-4
    # (f16x3 / exact fp32) against the reference's 3.99e-3: the product is SIX TIMES CLOSER to the exact gradient than the reference's own
    # fp32 run, i.e. the 4e-3 between product and reference (the round-5 question) is the reference's error, not the product's.  d delta:
    # transformer block 9.2e-5 against the reference's 3.85e-4; ResBlock (four numbers) 3.1e-5 against 7.2e-6 -- between the reference's
    # responses to a 1-ulp (1.0e-5) and a 32-ulp (9.9e-5) input perturbation measured against the same exact value, and unchanged when the
    # partial sums of d delta are kept in double (tried): per-element rounding of the chain, not the reduction.
    # Gates: d alpha at 2x the reference's own distance to the exact value; d delta at the larger of that and the reference's 32-ulp response.
    G20_FP64_FACTOR = 2.0
    u32 = golden("g20_reference_ulp32")
    resp32_a = rel(u32["grad0/%s/a" % name].astype(np.float64), ta64)
    for mode, r in runs.items():
        assert r["gw64"] <= G20_FP64_FACTOR * ref64[0], (mode, r["gw64"], ref64)
        assert r["ga64"] <= max(G20_FP64_FACTOR * ref64[1], resp32_a), (mode, r["ga64"], ref64, resp32_a)
        assert r["sgn64"] <= 2 * int((np.sign(rw8)[tw64 != 0] != np.sign(tw64)[tw64 != 0]).sum()) + 4, (mode, r["sgn64"])
    # ... and in index space against the EXACT run of all 12 iterations (g20_reference_fp64_full: the same graph, draws and masks in
    # float64): how many final hard roundings / first-step directions does the reference's fp32 run miss, and how many the product?
    f64f = golden("g20_reference_fp64_full")
    x_sign, x_near = _g20.unpack(f64f["final/%s/sign" % name], n), _g20.unpack(f64f["final/%s/near" % name], n)
    x_up, x_moved = _g20.unpack(f64f["first/%s/up" % name], n), _g20.unpack(f64f["first/%s/moved" % name], n)
    ref_dis64 = ref_sign != x_sign
    ref_far64 = int((ref_dis64 & ~(ref_near & x_near)).sum())
    ref_first64 = int(((ref_up != x_up) & (ref_moved | x_moved)).sum())
    print("G20 %s vs the EXACT 12-iteration run: reference fp32 -- first-step direction differs on %d, final hard rounding on %d (%d not next to zero in both)"
          % (name, ref_first64, int(ref_dis64.sum()), ref_far64))
    for mode, r in runs.items():
        tw = r["tw"]
        up, moved = (tw[0] > r["alpha0"]).cpu().numpy(), (tw[0] != r["alpha0"]).cpu().numpy()
        sign, near_got = (tw[-1] >= 0).cpu().numpy(), (tw[-1].abs() < _g20.NEAR).cpu().numpy()
        r["first64"] = int(((up != x_up) & (moved | x_moved)).sum())
        d64 = sign != x_sign
        r["dis64"], r["far64"] = int(d64.sum()), int((d64 & ~(near_got & x_near)).sum())
        print("G20 %s [%s] vs the EXACT 12-iteration run: first-step direction differs on %d, final hard rounding on %d (%d not next to zero in both)"
              % (name, "f16x3" if mode else "exact fp32", r["first64"], r["dis64"], r["far64"]))
        # What the tolerance of north_star is about -- the FINAL hard roundings -- is gated at the reference's own miss count against exact
        # arithmetic (x2, + a handful for the near-empty ResBlock counts).
        assert r["dis64"] <= 2 * int(ref_dis64.sum()) + 64 and r["far64"] <= 2 * ref_far64 + 8, \
            (mode, r["dis64"], r["far64"], int(ref_dis64.sum()), ref_far64)
        # The first Adam step is another matter: it moves alpha by lr g / (|g| + 1e-8), and an alpha whose gradient is ~1e-8 or less either
        # moves by less than its own rounding granularity or not at all -- whether (up, moved) agree there depends on the RELATIVE error of
        # gradient entries that are ten orders of magnitude below the layer's norm (cancellation residues of a 32768-term sum), which no
        # fp32 evaluation determines.  Measured: ResBlock 2505 such alphas (0.1 %; reference 40), and they carry a negligible share of the
        # gradient: gated on that share, and on 2 x the reference's count only where the count is not dominated by them.
        bad = ((up != x_up) & (moved | x_moved))
        gabs = np.abs(r["g0w"])
        share = float(gabs[bad].sum() / gabs.sum())
        print("   first-step disagreements carry %.2e of sum |d loss / d alpha| (median |g| among them %.2e, over all alphas %.2e)"
              % (share, float(np.median(gabs[bad])) if bad.any() else 0.0, float(np.median(gabs))))
        assert share <= 1e-3, (mode, share)
        assert r["first64"] <= max(2 * ref_first64 + 64, int(2e-3 * n)), (mode, r["first64"], ref_first64)
    # Measured (round 4, MI355X): ResBlock 192 -> 384 at 32 x 32, 2 359 296 alphas -- first-step direction 2491 (f16x3) / 2497 (exact fp32),
    # final rounding 71 / 65 (3 / 2 not next to zero), reference vs itself 2.  Transformer block d = 384 x 1024 tokens, 3 047 424
    # alphas -- first step 6941 / 6704, final 8788 / 8660 (2110 / 2056 not next to zero), reference vs itself 534: its softmax
    # step sizes (~0.004) move 2.5 % per Adam step at the shipped lr_a = 1e-4, and a noise-level gradient decides the direction.
    # Gates at 2x the measurement.  The production arithmetic (three f16 products) and the exact-fp32 MFMA are equally far from
    # the reference: what separates the GPU from the CPU is the order of its fp32 sums, not the operand expansion.
    # Round 6: the bounds against the REFERENCE are no longer "2x what we saw": the set of alphas on which the product and the reference
    # disagree is contained in (product vs exact) + (reference vs exact), so with the product gated against the exact run above
    # (<= 2 x the reference's own misses + slack) the disagreement with the reference is at most 3 x the reference's misses + slack.
    for mode, r in runs.items():
        assert len(r["dis"]) <= r["dis64"] + int(ref_dis64.sum()) <= 3 * int(ref_dis64.sum()) + 64, (mode, len(r["dis"]), r["dis64"], int(ref_dis64.sum()))
        assert r["first_bad"] <= r["first64"] + ref_first64, (mode, r["first_bad"], r["first64"], ref_first64)
    # the three-product contraction must not be a worse citizen than the exact-fp32 one
    assert len(bad[True]) <= 1.25 * len(bad[False]) + 8, (len(bad[True]), len(bad[False]))
    assert runs[True]["first_bad"] <= 1.25 * runs[False]["first_bad"] + 8


# ---------------------------------------------------------------------------------------------------------------------------
# configs 2, 3, 5 at full size: the three checks of the LDM-4 test above
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,rows", [("cifar", 64), ("church", 16), ("sd", 4)])
def test_other_configs_engine_matches_fake_quant_graph_at_full_size(kind, rows):
    """BASELINE configs 2 (CIFAR-10 DDPM UNet, configs/cifar10.yml:12-24: ch 128, mult 1-2-2-2, 35.7 M parameters), 3 (LSUN-Church
    LDM-8, models/ldm/lsun_churches256/config.yaml:32-53: ch 192, mult 1-2-2-4-4, legacy 8-head attention at every level, 295 M) and 5
    (Stable Diffusion v1-4, configs/stable-diffusion/v1-inference.yaml: ch 320, mult 1-2-4-4, 8 heads, 77 x 768 context, 860 M) at
    FULL size, random-init weights, W4A8 scales from the build's own quick initialisation (tools/config_bench.py::build): the frozen
    int8 engine against the fake-quant module graph (the calibration-time forward pinned to the reference on the fixture nets), replay
    bit-identity, row independence."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config_bench as cb
    dev = torch.device("cuda", 0)
    with torch.no_grad():
        qnn, inputs, _, _ = cb.build(kind, dev)
        x, t, c = inputs(rows)
        qnn.set_quant_state(True, True)
        fq = qnn(x, t, c).float()
        eng = qnn.freeze()
        out = qnn(x, t, c).float()
        assert qnn.engine is not None
        modes = cb._modes(eng)
        rng = float(fq.abs().max())
        err = (out - fq).abs()
        print("full-size %s, %d rows: engine vs fake-quant graph max err %.3e of range, mean %.3e of range; layer modes %s"
              % (kind, rows, float(err.max()) / rng, float(err.mean()) / rng, modes))
        assert modes.get("i8", 0) >= 50
        assert torch.isfinite(out).all()
        # same bound as the headline network: an activation within ~1e-6 of a rounding boundary flips one code, and 100+ quantised
        # layers deep the flips accumulate
        assert float(err.max()) <= 0.10 * rng and float(err.mean()) <= 0.01 * rng
        assert torch.equal(qnn(x, t, c).float(), out)                                   # replay: bit-identical
        perm = torch.arange(rows - 1, -1, -1, device=dev)                               # row independence
        out_p = qnn(x[perm], t[perm], None if c is None else c[perm]).float()
        assert float((out_p[perm] - out).abs().max()) == 0.0
