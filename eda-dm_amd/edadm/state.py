"""Save / load of the calibrated quantisation state.  The reference never persists it (a
multi-hour calibration is lost with the process); this build defines a flat
name -> array format so calibration and multi-GPU sampling can run as separate jobs:

    <module path>/delta, <module path>/zero_point, <module path>/n_bits   every quantizer
    <module path>/alpha                                                   AdaRound quantizers
    <module path>/split                                                   QuantModules with a channel split

Module paths are those of `QuantModel.named_modules()` (identical to the reference's).

A second file holds the *frozen integer model* the sampling jobs load (`save_frozen` / `load_frozen`): per layer the
folded bias, activation-quantiser table, per-channel scales and the hard-rounded integer weights, 4-bit layers as
packed nibbles (`Engine.export_frozen`, edadm_pack_w4 / edadm_unpack_w4) -- 0.5 byte per weight, ~200 MB for LDM-4."""
import numpy as np
import torch

from qdiff.quant_layer import UniformAffineQuantizer, QuantModule
from qdiff.adaptive_rounding import AdaRoundQuantizer


def quant_state_dict(qnn):
    out = {}
    for name, m in qnn.named_modules():
        if isinstance(m, QuantModule) and m.split:
            out[name + "/split"] = np.int64(m.split)
        if isinstance(m, (UniformAffineQuantizer, AdaRoundQuantizer)) and m.delta is not None:
            out[name + "/delta"] = m.delta.detach().cpu().numpy()
            out[name + "/zero_point"] = m.zero_point.detach().cpu().numpy()
            out[name + "/n_bits"] = np.int64(m.n_bits)
            if isinstance(m, AdaRoundQuantizer):
                out[name + "/alpha"] = m.alpha.detach().cpu().numpy()
    return out


def load_quant_state(qnn, state, prefix=""):
    """Restore deltas / zero points / bit widths (and splits) captured by `quant_state_dict` or by
    the reference (golden fixtures use prefix 'qp/').  Returns the number of quantizers restored."""
    dev = next(qnn.parameters()).device
    for name, m in qnn.named_modules():
        if isinstance(m, QuantModule):
            k = prefix + name + "/split"
            split = int(state[k]) if k in state else m.split
            if split and not m.split:
                m.split = split
                m.set_split()
                m.to(dev)
    n = 0
    for name, m in qnn.named_modules():
        if not isinstance(m, UniformAffineQuantizer):
            continue
        k = prefix + name
        if k + "/delta" not in state:
            continue
        delta = torch.as_tensor(np.asarray(state[k + "/delta"]), dtype=torch.float32, device=dev)
        m.zero_point = torch.as_tensor(np.asarray(state[k + "/zero_point"]), dtype=torch.float32, device=dev)
        m.bitwidth_refactor(int(state[k + "/n_bits"]))
        m.delta = torch.nn.Parameter(delta) if m.leaf_param else delta
        m.set_inited(True)
        n += 1
    return n


def save_frozen(qnn, path):
    """Freeze `qnn` (if it is not yet) and write the integer model to `path` (.npz).  Returns bytes written."""
    import os
    eng = qnn.engine if getattr(qnn, "engine", None) is not None else qnn.freeze()
    np.savez(path, **eng.export_frozen())
    return os.path.getsize(path if str(path).endswith(".npz") else str(path) + ".npz")


def load_frozen(qnn, path):
    """Build the engine of `qnn` (same topology, quantiser state already loaded with `load_quant_state`) and replace
    its integer weights by the saved ones.  Returns the engine."""
    eng = qnn.engine if getattr(qnn, "engine", None) is not None else qnn.freeze()
    with np.load(path, allow_pickle=False) as z:
        eng.load_frozen({k: z[k] for k in z.files})
    return eng
