"""Round-4 oracle pins (CPU): fixture G20 -- the reference's reconstruction of ONE LDM-4-sized ResBlock (192 -> 384 at 32 x 32, 32-row
minibatches, shipped ImageNet hyper-parameters, 0.5 / 0.5 masks; tests/golden/_g20.py, make_golden.py::g20_f16x3_units over
qdiff_control/block_recon.py:13-243).  The CPU restatement replays its first iterations from the same formulas and must land
on the reference's alphas; the counter-based uniform generator the large fixtures use is checked in its numpy and torch forms."""
import os
import random
import sys

import numpy as np
import torch

from oracle import qdiff_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import _g20  # noqa: E402
import _uniforms  # noqa: E402
from _weights import formula_state_dict  # noqa: E402

WQ4 = dict(n_bits=4, symmetric=True, channel_wise=True, scale_method="mse")
AQ8 = dict(n_bits=8, symmetric=True, channel_wise=False, scale_method="mse", leaf_param=True, prob=0.5)


def test_hash_uniforms_numpy_and_torch_agree_and_match_the_fixture(golden):
    g = golden("g20_f16x3_units")
    for line, chk in list(zip(g["rand/log"], g["rand/check"]))[:40]:
        owner, phase, c, shp = str(line).split("|")
        shape = [int(s) for s in shp.split("x")]
        if int(np.prod(shape)) > 8_000_000:
            continue
        u = _uniforms.uniform_hash(owner, phase, int(c), shape)
        assert float(u.reshape(-1)[0]) == chk[0] and float(u.reshape(-1)[-1]) == chk[1]
        assert abs(float(u.astype(np.float64).sum()) - chk[2]) < 1e-6
        t = _uniforms.uniform_hash_torch(owner, phase, int(c), shape, "cpu").numpy()
        assert np.array_equal(u, t)
        assert u.min() >= 0.0 and u.max() < 1.0


def _res_unit(g):
    shapes = {"in_layers.0.weight": (192,), "in_layers.0.bias": (192,), "in_layers.2.weight": (384, 192, 3, 3), "in_layers.2.bias": (384,),
              "emb_layers.1.weight": (384, 768), "emb_layers.1.bias": (384,), "out_layers.0.weight": (384,), "out_layers.0.bias": (384,),
              "out_layers.3.weight": (384, 384, 3, 3), "out_layers.3.bias": (384,), "skip_connection.weight": (384, 192, 1, 1),
              "skip_connection.bias": (384,)}
    sd = formula_state_dict([("res." + k, v) for k, v in shapes.items()], _g20.SEED)
    B = O._Builder({k: torch.as_tensor(v) for k, v in sd.items()}, WQ4, AQ8, 8)
    return O.OResBlock(B, "res", "res", 192, 384)


def test_g20_resblock_first_iterations_on_the_cpu_restatement(golden):
    g = golden("g20_f16x3_units")
    unit = _res_unit(g)
    (xq, eq), (xf, ef) = _g20.caches("res")
    xq, eq, xf, ef = (torch.from_numpy(a) for a in (xq, eq, xf, ef))
    # FP targets: the restatement's own FP forward reproduces the reference's samples
    unit.set_quant_state(False, False)
    with torch.no_grad():
        out_fp = torch.cat([unit(xf[i:i + 32], ef[i:i + 32]) for i in range(0, _g20.ROWS, 32)])
    pos = torch.from_numpy(_g20.sample_positions(out_fp.numel()))
    ref = torch.from_numpy(g["out_fp/res/sample"])
    assert float((out_fp.reshape(-1)[pos] - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    # the reference's initial scales
    n = 0
    for l in unit.layers():
        for q in l.quantizers():
            k = "init/qp/model." + q.name
            q.delta, q.zero_point = torch.from_numpy(g[k + "/delta"]).float(), torch.from_numpy(g[k + "/zero_point"]).float()
            q.bitwidth_refactor(int(g[k + "/n_bits"]))
            q.inited = True
            n += 1
    assert n == len([k for k in g.files if k.startswith("init/qp/model.res.") and k.endswith("/delta")])
    rep = _uniforms.ReplayHash()
    for l in unit.layers():
        l.act_quantizer.mask_fn = (lambda nm: lambda xx: torch.from_numpy(rep.draw("model." + nm, "iter", xx.shape)))(l.act_quantizer.name)
    steps = 2
    tw = []

    class _Stop(Exception):
        pass

    def trace(it, wp, ap, loss):
        tw.append(torch.cat([p.detach().flatten() for p in wp]).clone())
        if len(tw) == steps:
            raise _Stop

    h = _g20.HYPER
    random.seed(_g20.SEED + 1)
    try:
        O.reconstruct_unit(None, unit, "block", cali=None, iters=int(g["iters"]), act_quant=True, lr_a=h["lr_a"], lr_w=h["lr_w"], p=2.0,
                           batch_size=h["batch_size"], input_prob=h["input_prob"], add_loss=h["add_loss"], recon_w=True, recon_a=True,
                           caches=(True, (xq, eq), (xf, ef), out_fp),
                           rand_fn=lambda xx: torch.from_numpy(rep.draw("input_mix:res", "iter", xx.shape)), trace=trace)
    except _Stop:
        pass
    ncount = int(g["final/res/count"])
    assert tw[0].numel() == ncount
    # the first Adam step is +- lr_w by the sign of the first gradient: compare directions with the reference's, then the strided
    # trajectory of both steps
    a0 = []
    for l in unit.layers():
        wq = l.weight_quantizer
        a0.append(O.adaround_init_alpha(l.weight, wq.delta).flatten())
    a0 = torch.cat(a0)
    up = (tw[0] > a0).numpy()
    ref_up = _g20.unpack(g["first/res/up"], ncount)
    ref_moved = _g20.unpack(g["first/res/moved"], ncount)
    moved = (tw[0] != a0).numpy()
    bad = int(((up != ref_up) & (moved | ref_moved)).sum())
    ref_w = g["traj/res/w"][:steps]
    got = torch.stack(tw)[:, ::_g20.STRIDE].numpy()
    dw = np.abs(got - ref_w)
    print("G20 res on the CPU restatement: first-step direction differs on %d of %d alphas; strided trajectory (2 steps) median %.2e, frac > lr/10 %.5f"
          % (bad, ncount, np.median(dw), (dw > 0.05).mean()))
    # measured: 32 directions of 2.36 M (the restatement's closed-form gradients against autograd on the same CPU kernels)
    assert bad <= 200 and np.median(dw) < 1e-4 and (dw > 0.05).mean() < 1e-3
