"""-m gpu: the reciprocal-multiply quantisers ON THE DEVICE at their rounding boundaries (round-4 review, item 2b).

tests/test_rounding_band.py restates `common.h rint_div*` in numpy; here the kernels themselves are fed adversarial values
(k + 1/2) delta +- {0..6} ulp -- a value the fast path t = v * (1 / delta) may round to the other integer than the reference's IEEE
division round(v / delta) (quant_layer.py:266-270) unless the band sends it to the exact path -- and every code is compared with the
IEEE division evaluated in numpy float32 on the host: ZERO mismatches, through
  * edadm_quant_i8 / edadm_quant_f16 (the stand-alone activation quantisers),
  * the quantising GEMM epilogues of edadm_qgemm_i8_q (int8 and f16 codes, with and without the fp32 residual; accumulators and
    per-column scales constructed so that the fp32 value in front of the quantiser is known exactly),
  * the softmax coders (edadm_softmax_quant_f16 and the fused attention kernels: rows of n equal scores give p = 1 / n exactly, and the
    step size is placed so that p / delta sits on a boundary).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
F = np.float32


def _walk(v, ulps):
    """v moved by `ulps` (array of small signed integers) representable steps"""
    v = v.astype(F).copy()
    for _ in range(int(np.abs(ulps).max())):
        s = np.sign(ulps)
        mv = s != 0
        v[mv] = np.nextafter(v[mv], np.where(s[mv] > 0, F(np.inf), F(-np.inf)).astype(F))
        ulps = ulps - s
    return v


def _adversarial(rng, n, d, kmax):
    """n values a few ulp around (k + 1/2) d, |k| < kmax, interleaved with plain ones"""
    k = rng.integers(-kmax, kmax, n).astype(np.float64)
    v = ((k + 0.5) * np.float64(d)).astype(F)
    v = _walk(v, rng.integers(-6, 7, n))
    plain = (rng.standard_normal(n) * 0.3 * kmax).astype(F) * F(d)
    out = np.empty(2 * n, F)
    out[0::2], out[1::2] = v, plain
    return out


def _ref_codes(v, d, z, qmax):
    """quant_layer.py:266-270 in IEEE float32: clamp(round(v / d) + z, 0, qmax)"""
    return np.clip(np.rint((v.astype(F) / F(d)).astype(F)) + F(z), 0, qmax).astype(F)


DELTAS = [0.047, 0.0123, 1.0 / 3.0, 0.5, 0.0009765625, 0.731, 2.5e-3, 0.09]


def test_quant_i8_and_f16_on_boundary_values():
    from edadm import ops
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    n_adv = 0
    for i, d in enumerate(DELTAS + [float(rng.uniform(1e-3, 1.0)) for _ in range(24)]):
        d = float(F(d))
        z = float(rng.integers(0, 256)) if i % 3 else 128.0
        v = _adversarial(rng, 1 << 18, d, 200).reshape(-1, 512)
        ref = _ref_codes(v, d, z, 255.0)
        qp = ops.qp_tensor([(d, z, 255.0)], dev)
        x = torch.from_numpy(v).to(dev)
        got8 = ops.quant_i8(x, qp).cpu().numpy().astype(F) + 128.0
        assert np.array_equal(got8, ref), (d, z, int((got8 != ref).sum()))
        got16 = ops.quant_f16(x, qp).float().cpu().numpy() + F(z)
        assert np.array_equal(got16, ref), (d, z, int((got16 != ref).sum()))
        n_adv += v.size // 2
    print("quant_i8 / quant_f16: %d boundary values (+ as many plain), 0 codes off" % n_adv)


def _gemm_case(rng, M, N, K, d, z, mode, with_res, dev):
    """int8 operands whose accumulators are known integers, per-column scales that put fl(acc * s_c) a few ulp around a .5 boundary
    of the output quantiser for the rows holding a = 127 (bias 0: the value in front of the quantiser is ONE rounding of an exact
    product, reproduced on the host through float64)."""
    from edadm import ops
    a_r = rng.integers(-127, 128, M).astype(np.int8)
    a_r[::2] = 127                                          # the adversarial rows
    w_c = rng.integers(1, 8, N).astype(np.int8) * rng.choice([-1, 1], N).astype(np.int8)     # 4-bit weights
    A = np.zeros((M, K), np.int8)
    A[:, 0] = a_r
    A[:, 1:] = rng.integers(-128, 128, (M, K - 1))
    W = np.zeros((N, K), np.int8)
    W[:, 0] = w_c                                           # only k = 0 contributes: acc[r][c] = a_r * w_c
    acc = a_r.astype(np.float64)[:, None] * w_c.astype(np.float64)[None, :]
    k_c = rng.integers(-100, 100, N).astype(np.float64)
    s = ((k_c + 0.5) * np.float64(d) / (127.0 * w_c.astype(np.float64))).astype(F)
    s = np.abs(_walk(s, rng.integers(-6, 7, N))).astype(F)
    s[s == 0] = F(d)
    v = (acc * s.astype(np.float64)[None, :]).astype(F)     # fma(acc, s, 0): one rounding of the exact product
    res = None
    if with_res:
        # whole steps of the output quantiser: the sum stays next to a boundary (float64 holds the sum of two float32 exactly)
        res = (rng.integers(-8, 9, (M, N)).astype(np.float64) * np.float64(d)).astype(F)
        v = (v.astype(np.float64) + res.astype(np.float64)).astype(F)
    ref = _ref_codes(v, d, z, 255.0)
    qp = ops.qp_tensor([(d, z, 255.0)], dev)
    out = ops.qgemm_i8_q(torch.from_numpy(A).to(dev), torch.from_numpy(W).to(dev), M, N, K, torch.from_numpy(s).to(dev),
                         torch.zeros(N, device=dev), mode, qp, residual=None if res is None else torch.from_numpy(res).to(dev))
    got = out.float().cpu().numpy() + (F(z) if mode == 1 else F(128.0))
    return got, ref, v


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("with_res", [False, True])
def test_quantising_gemm_epilogues_on_boundary_values(mode, with_res):
    """edadm_qgemm_i8_q out_mode 1 (f16 code - zp) / 2 (int8 code - 128): full tiles (register-direct epilogue: the persistent
    4-wave kernel at >= 512 tiles, the per-tile one below) and a ragged shape (LDS-staged epilogue)."""
    from edadm import lib
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(100 + 10 * mode + int(with_res))
    total = adv = 0
    for (M, N, K) in ((65536, 384, 64), (4096, 576, 128), (1000, 192, 64)):
        for d in (0.047, 0.0123, 1.0 / 3.0, float(rng.uniform(1e-2, 1.0))):
            d = float(F(d))
            z = float(rng.integers(0, 256))
            got, ref, v = _gemm_case(rng, M, N, K, d, z, mode, with_res, dev)
            bad = got != ref
            assert not bad.any(), (M, N, K, d, z, int(bad.sum()), v[bad][:4], got[bad][:4], ref[bad][:4])
            total += got.size
            t = (v / F(d)).astype(F)
            adv += int((np.abs(t - np.rint(t)) > 0.499).sum())
    print("qgemm_i8_q mode %d residual %s: %d codes, %d of them within 1e-3 of a boundary, 0 off" % (mode, with_res, total, adv))
    assert adv > 1e5


def test_softmax_coder_on_boundary_probabilities():
    """rows of n equal scores (the rest -1e30): every numerator is exp(0) = 1, the sum is n, p = fl(1 / n); delta is placed so that
    p / delta is a few ulp around k + 1/2.  Reference: fl(fl(1 / n) / delta), quant_block.py:128-162 on torch's softmax."""
    from edadm import ops
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    cols, checked = 512, 0
    ns = np.array([1, 2, 3, 5, 7, 10, 37, 64, 100, 129, 255, 300, 511, 512])
    for k in (0, 1, 2, 7, 30, 100, 200):
        for n in ns:
            p = (F(1.0) / F(n)).astype(F)
            d0 = np.array([np.float64(p) / (k + 0.5)]).astype(F)
            for j in range(-6, 7):
                d = float(_walk(d0, np.array([j]))[0])
                s = np.full((64, cols), -1e30, F)
                s[:, :n] = F(rng.uniform(-3, 3))             # any common score: the maximum is subtracted
                qp = ops.qp_tensor([(d, 0.0, 255.0)], dev)
                got = ops.softmax_quant_f16(torch.from_numpy(s).to(dev), qp).float().cpu().numpy()
                ref = min(float(np.rint((p / F(d)).astype(F))), 255.0)
                assert (got[:, :n] == ref).all() and (got[:, n:] == 0).all(), (n, k, j, d, got[0, :2], ref)
                checked += 1
    print("softmax_quant_f16: %d (n, k, ulp) boundary cases, 0 codes off" % checked)


@pytest.mark.parametrize("i8", [False, True])
def test_fused_attention_coder_on_boundary_probabilities(i8):
    """The fused attention kernels code their probabilities in registers (csrc/attn.hip).  q = 0 makes every score 0: p = 1 / Nk exactly;
    with v = 1 the output is Nk * code * alpha_pv, which reads the code back.  d = 384, Nk = 1024: the wide-head kernels
    (k_attn_wide16 / k_attn_wide16_i8) of the headline's 32 x 32 level; d = 64, Nk = 256: the general fused kernel."""
    from edadm import ops
    dev = torch.device("cuda", 0)
    checked = 0
    for (d_head, N) in ((384, 1024), (64, 256)):
        if i8 and not ops.attention_i8qk_ok(1, d_head, N, N):
            continue
        if not i8 and not ops.attention_fused_ok(1, d_head, N, N):
            continue
        p = (F(1.0) / F(N)).astype(F)
        B = 2
        v = torch.ones(B * N, d_head, dtype=torch.float16, device=dev)
        for k in (0, 1, 3, 20, 127, 254):
            d0 = np.array([np.float64(p) / (k + 0.5)]).astype(F)
            for j in range(-6, 7):
                dl = float(_walk(d0, np.array([j]))[0])
                qp = ops.qp_tensor([(dl, 0.0, 255.0)], dev)
                if i8:
                    q8 = torch.full((B * N, d_head), -128, dtype=torch.int8, device=dev)     # code 0 with zero point 0: q - zq = 0
                    k8 = torch.randint(-128, 128, (B * N, d_head), dtype=torch.int8, device=dev)
                    out = ops.attention_fused_i8qk(q8, k8, v, B, 1, N, N, d_head, 0.01, 0.0, qp, 1.0)
                else:
                    q = torch.zeros(B * N, d_head, dtype=torch.float16, device=dev)
                    kk = torch.randint(-100, 100, (B * N, d_head), device=dev).half()
                    out = ops.attention_fused(q, kk, v, B, 1, N, N, d_head, 0.01, qp, 1.0)
                ref = min(float(np.rint((p / F(dl)).astype(F))), 255.0)
                got = out.float().cpu().numpy() / N
                assert (got == ref).all(), (d_head, N, k, j, dl, float(got.min()), float(got.max()), ref)
                checked += 1
    print("fused attention (%s scores): %d boundary cases, 0 codes off" % ("int8" if i8 else "f16", checked))
    assert checked > 0
