"""Save / load of the calibrated quantisation state.  The reference never persists it (a
multi-hour calibration is lost with the process); this build defines a flat
name -> array format so calibration and multi-GPU sampling can run as separate jobs:

    <module path>/delta, <module path>/zero_point, <module path>/n_bits   every quantizer
    <module path>/alpha                                                   AdaRound quantizers
    <module path>/split                                                   QuantModules with a channel split

Module paths are those of `QuantModel.named_modules()` (identical to the reference's)."""
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
