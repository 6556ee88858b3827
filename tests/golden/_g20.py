"""G20: two reconstruction units at a size where the product's three-product f16 contraction is the arithmetic that runs
(edadm/contract.py: _f16x3_linear, f16x3_conv_ok, the direct 3x3 kernel) -- the LDM-4 ImageNet ResBlock 192 -> 384 at 32 x 32 and
its transformer block (one head, d = 384, 1024 tokens, one-token context of 512), 32-row minibatches over 64 cached rows.
Nothing of that size can be stored: weights (tests/golden/_weights.py), cached unit inputs (below) and stochastic masks
(tests/golden/_uniforms.py: uniform_hash) are pure functions of (seed, name); the fixture stores what the REFERENCE made of
them (tests/golden/make_golden.py::g20_f16x3_units).  Shared by the generator, the CPU oracle test and the -m gpu test."""
import zlib

import numpy as np

SEED = 2020
ROWS, BATCH, ITERS = 64, 32, 12
RES = dict(channels=192, emb_channels=768, out_channels=384, hw=32)
TF = dict(dim=384, heads=1, d_head=384, context_dim=512, tokens=1024)
# the shipped ImageNet setting (scripts/for_imagenet.sh:16, sample_diffusion_ldm_imagenet.py:144,178-195)
HYPER = dict(act_quant=True, asym=True, opt_mode="mse", lr_a=1e-4, lr_w=5e-1, p=2.0, weight=0.0001, b_range=(20, 2), warmup=0.2,
             batch_size=BATCH, input_prob=0.5, add_loss=0.8, recon_w=True, recon_a=True, keep_gpu=True)
PROB = 0.5
STRIDE = 61            # trajectories are stored at every STRIDE-th trainable
NEAR = 0.05            # |alpha| below a tenth of the first Adam step: "next to the rounding boundary"


def normal(name, shape, scale=1.0):
    rs = np.random.RandomState((SEED * 1000003 + zlib.crc32(name.encode())) % (2 ** 32))
    return (rs.standard_normal(tuple(shape)) * scale).astype(np.float32)


def caches(unit):
    """(inp_q, second_q), (inp_fp, second_fp): the unit inputs a quantised / a full-precision prefix would have produced.  The
    quantised-prefix version deviates by a few per cent, like the cached tensors of a real walk; the second input is the time
    embedding (ResBlock) or the context (transformer block: never quantised upstream, the same tensor in both)."""
    if unit == "res":
        x = normal("res/x", (ROWS, RES["channels"], RES["hw"], RES["hw"]))
        e = normal("res/emb", (ROWS, RES["emb_channels"]))
        return (x + normal("res/dx", x.shape, 0.05), e + normal("res/demb", e.shape, 0.02)), (x, e)
    x = normal("tf/x", (ROWS, TF["tokens"], TF["dim"]))
    c = normal("tf/ctx", (ROWS, 1, TF["context_dim"]))
    return (x + normal("tf/dx", x.shape, 0.05), c), (x, c)


def sample_positions(n, count=4096):
    """positions of the stored samples of a tensor with n elements"""
    return (np.arange(count, dtype=np.int64) * 2654435761 + 12345) % n


def pack(bits):
    return np.packbits(np.asarray(bits).reshape(-1).astype(np.uint8))


def unpack(packed, n):
    return np.unpackbits(packed)[:n].astype(bool)
