"""Formula weights for the larger fixtures: every parameter is a pure function of (seed, name, shape) through numpy's
legacy MT19937 stream (stable across numpy versions), so a fixture stores inputs, outputs and quantiser state but not
the megabytes of weights -- the generator (make_golden.py, reference classes) and the tests (product classes) both
rebuild them from this one function."""
import zlib

import numpy as np


def formula_state_dict(named_shapes, seed):
    """named_shapes: iterable of (state_dict key, shape).  Matrices / filters ~ N(0, 1/fan_in), 1-D `weight`
    (normalisation gains) ~ 1 + 0.1 N, biases ~ 0.05 N."""
    out = {}
    for name, shape in named_shapes:
        shape = tuple(int(s) for s in shape)
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 32))
        a = rs.standard_normal(shape).astype(np.float32)
        if name.endswith("weight") and len(shape) >= 2:
            a *= np.float32(1.0 / np.sqrt(np.prod(shape[1:])))
        elif name.endswith("weight"):
            a = (1.0 + 0.1 * a).astype(np.float32)
        else:
            a *= np.float32(0.05)
        out[name] = a
    return out
