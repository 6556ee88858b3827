"""Uniform draws for the stochastic-mask fixtures (G8b, G18): the reference calls `torch.rand_like` for the input mix
(block_recon.py:141-145) and inside every training-mode activation quantizer (quant_layer.py:271-275).  The fixture generator
patches `torch.rand_like` with `uniform(owner, phase, count, shape)` below -- a pure function of (who asked, in which phase,
how many times before) -- and records every call; the tests inject the same tensors into the oracle / the product, so the
masks the reference consumed are replayed exactly without storing megabytes of random numbers.  The fixture also stores
(first values, sum) of every call so that a drift of the generator under another torch build is detected, not absorbed."""
import zlib

import numpy as np


def uniform(owner, phase, count, shape):
    """numpy legacy MT19937 (stable across numpy versions) -> float32 in [0, 1)."""
    seed = zlib.crc32(("%s|%s|%d" % (owner, phase, count)).encode()) % (2 ** 32)
    rs = np.random.RandomState(seed)
    n = int(np.prod(shape))
    # 24 random bits -> exactly representable float32 in [0, 1), like torch.rand
    return (rs.randint(0, 1 << 24, size=n).astype(np.float32) * np.float32(2.0 ** -24)).reshape(tuple(shape))


class Replay:
    """Per-owner call counters for the test side: `draw(owner, phase, shape)`."""

    def __init__(self):
        self.counts = {}
        self.log = []

    def draw(self, owner, phase, shape):
        key = (owner, phase)
        c = self.counts.get(key, 0)
        self.counts[key] = c + 1
        self.log.append((owner, phase, c, tuple(int(s) for s in shape)))
        return uniform(owner, phase, c, shape)

    def draw_calls(self, owner, phase, shape, batch):
        """A product call over the batch [x | x] (edadm/recon.py BATCH_FORWARDS: the two quantised forwards of a block iteration
        as one) stands for TWO calls of the reference, in its order: the first half takes the earlier draw.  batch: truthy when the
        unit runs batched (edadm.recon.STATE["batched"])."""
        shape = tuple(int(s) for s in shape)
        if batch and shape[0] % 2 == 0 and batch > 0:
            half = (shape[0] // 2,) + shape[1:]
            return np.concatenate([self.draw(owner, phase, half), self.draw(owner, phase, half)])
        return self.draw(owner, phase, shape)
