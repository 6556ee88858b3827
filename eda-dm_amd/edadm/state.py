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
    if hasattr(qnn, "engine"):
        qnn.engine = None        # compiled from the previous state: freeze() / load_frozen() again
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
            continue                                   # AdaRound quantizers: below
        k = prefix + name
        if k + "/delta" not in state:
            continue
        delta = torch.as_tensor(np.asarray(state[k + "/delta"]), dtype=torch.float32, device=dev)
        m.zero_point = torch.as_tensor(np.asarray(state[k + "/zero_point"]), dtype=torch.float32, device=dev)
        m.bitwidth_refactor(int(state[k + "/n_bits"]))
        m.delta = torch.nn.Parameter(delta) if m.leaf_param else delta
        m.set_inited(True)
        n += 1
    # learned rounding: a weight quantizer saved as an AdaRoundQuantizer comes back as one (hard mode), its alpha restored,
    # so the quantiser-state file alone reproduces the calibrated integer weights
    for name, m in qnn.named_modules():
        if not isinstance(m, QuantModule):
            continue
        for attr in ("weight_quantizer", "weight_quantizer_0"):
            k = "%s%s.%s/alpha" % (prefix, name, attr)
            q = getattr(m, attr, None)
            if k not in state or q is None:
                continue
            alpha = torch.as_tensor(np.asarray(state[k]), dtype=torch.float32, device=dev)
            if not isinstance(q, AdaRoundQuantizer):
                w = m.org_weight.data
                if m.split:
                    w = w[:, :m.split, ...] if attr == "weight_quantizer" else w[:, m.split:, ...]
                q = AdaRoundQuantizer(uaq=q, round_mode='learned_hard_sigmoid', weight_tensor=w)
                setattr(m, attr, q)
            else:
                q.delta, q.zero_point = (torch.as_tensor(np.asarray(state[k[:-5] + sfx]), dtype=torch.float32, device=dev)
                                         for sfx in ("delta", "zero_point"))
                q._d, q._z = q.delta.reshape(-1).contiguous(), q.zero_point.reshape(-1).contiguous()
            assert tuple(alpha.shape) == tuple(q.alpha.shape), (k, alpha.shape, q.alpha.shape)
            with torch.no_grad():
                q.alpha.copy_(alpha)
            q.soft_targets = False
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
