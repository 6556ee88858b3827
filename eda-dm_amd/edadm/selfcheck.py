"""smoke(): one small invocation of the hot path on cuda:0 — the tiny LDM-4-shaped fixture network
through the frozen int8 executor — checked against the oracle run with the same parameters."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_smoke():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import build_ldm, quantize_like_reference, WQ4, AQ8
    from oracle import qdiff_oracle as O           # checker only
    g = np.load(os.path.join(ROOT, "tests", "golden", "g13_ldm_imagenet.npz"))
    model = build_ldm(g)
    qnn, (x, t, ctx), n = quantize_like_reference(model, g, "ldm")
    qnn.set_quant_state(True, True)
    eng = qnn.freeze()
    with torch.no_grad():
        out = qnn(x, t, ctx).cpu().numpy()
    cfg = {k[4:]: g[k] for k in g.files if k.startswith("cfg/")}
    net = O.OUNet({k[3:]: g[k] for k in g.files if k.startswith("sd/")}, WQ4, AQ8, 8, **cfg)
    net.set_first_last_layer_to_8bit()
    net.disable_network_output_quantization()
    net.split_shortcut = True
    with torch.no_grad():
        net(x.cpu(), t.cpu(), ctx.cpu())        # FP pass creates the split quantizers
    net.load_qparams(g)
    net.set_quant_state(True, True)
    with torch.no_grad():
        ref = net(x.cpu(), t.cpu(), ctx.cpu()).numpy()
    rng = np.abs(ref).max()
    err = np.abs(out - ref)
    modes = {}
    for L in eng.layers.values():
        modes[L.mode] = modes.get(L.mode, 0) + 1
    print("smoke: int8 engine vs oracle: max err %.3e of range, mean %.3e of range; layers %s"
          % (err.max() / rng, err.mean() / rng, modes))
    assert err.max() <= 5e-2 * rng and err.mean() <= 5e-3 * rng
    assert modes.get("i8", 0) >= 100
