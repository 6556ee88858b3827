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


# ---- large fixtures (G20): a counter-based generator that numpy (fixture generation, CPU oracle) and torch on the GPU
# (the -m gpu test: no host generation, no upload of hundreds of megabytes per iteration) evaluate to the same bits ----
_M1, _M2, _GOLD = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB, 0x9E3779B97F4A7C15


def hash_seed(owner, phase, count):
    return zlib.crc32(("%s|%s|%d" % (owner, phase, count)).encode()) % (2 ** 32)


def uniform_hash(owner, phase, count, shape):
    """splitmix64 finaliser of (seed * golden + element index) -> top 24 bits -> float32 in [0, 1)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64((hash_seed(owner, phase, count) * _GOLD) % (1 << 64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(tuple(int(s) for s in shape))


def uniform_hash_torch(owner, phase, count, shape, device):
    """the same values from torch int64 arithmetic (wrapping multiply; logical shifts as arithmetic shift + mask)"""
    import torch

    def s64(v):                                   # two's-complement view of an unsigned 64-bit constant
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    n = int(np.prod(shape))
    z = torch.arange(n, dtype=torch.int64, device=device) + s64((hash_seed(owner, phase, count) * _GOLD) % (1 << 64))
    z = (z ^ lsr(z, 30)) * s64(_M1)
    z = (z ^ lsr(z, 27)) * s64(_M2)
    z = z ^ lsr(z, 31)
    return (lsr(z, 40).to(torch.float32) * (2.0 ** -24)).reshape(tuple(int(s) for s in shape))


class ReplayHash(Replay):
    """Replay whose draws come from uniform_hash (numpy) or, with a device, uniform_hash_torch."""

    def __init__(self, device=None):
        super().__init__()
        self.device = device

    def draw(self, owner, phase, shape):
        key = (owner, phase)
        c = self.counts.get(key, 0)
        self.counts[key] = c + 1
        self.log.append((owner, phase, c, tuple(int(s) for s in shape)))
        if self.device is None:
            return uniform_hash(owner, phase, c, shape)
        return uniform_hash_torch(owner, phase, c, shape, self.device)

    def draw_calls(self, owner, phase, shape, batch):
        shape = tuple(int(s) for s in shape)
        if batch and shape[0] % 2 == 0 and batch > 0:
            half = (shape[0] // 2,) + shape[1:]
            a, b = self.draw(owner, phase, half), self.draw(owner, phase, half)
            if self.device is None:
                return np.concatenate([a, b])
            import torch
            return torch.cat([a, b])
        return self.draw(owner, phase, shape)
