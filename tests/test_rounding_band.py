"""The reciprocal-multiply quantisers (csrc/common.h rint_div / rint_div_n / rint_div_zp_n, csrc/attn.hip softmax codes) round through
t = v * (1 / delta) and redo the IEEE division only inside a band around the .5 boundaries.  CPU restatement of that arithmetic in
numpy float32 (the FMA through float64: the product of two float32 is exact there) -- OUTSIDE the band the fast integer must equal the
reference's round(x / delta) (+ zero point) (quant_layer.py:266-276) for adversarial inputs placed a few ulp around every boundary;
the band must also be narrow enough to be rare on ordinary inputs (the exact path is a wave-wide branch)."""
import numpy as np

F = np.float32
BAND_REL, SM_BAND_REL = F(2.4e-7), F(3.6e-7)


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def _near_limit(z):
    return F(0.5) - _fma(np.abs(z), F(1.2e-7) * np.ones_like(z), F(4e-5) * np.ones_like(z))


def _adversarial(rng, n, kmax):
    """values a few ulp around (k + 0.5) * d, plus plain random ones"""
    d = np.exp(rng.uniform(np.log(1e-4), np.log(8.0), n)).astype(F)
    k = rng.integers(-kmax, kmax, n).astype(np.float64)
    v = ((k + 0.5) * d.astype(np.float64)).astype(F)
    ulps = rng.integers(-6, 7, n)
    for _ in range(6):                                       # walk up to 6 ulp either way
        step = np.sign(ulps).astype(F)
        move = ulps != 0
        v = np.where(move, np.nextafter(v, v + step * np.abs(v) - (step == 0)), v).astype(F)
        ulps = ulps - np.sign(ulps)
    plain = (rng.standard_normal(n) * 40).astype(F) * d
    return np.concatenate([v, plain]), np.concatenate([d, d])


def test_rint_div_fast_path_equals_the_division_outside_the_band():
    rng = np.random.default_rng(7)
    v, d = _adversarial(rng, 2_000_000, 300)
    inv = (F(1.0) / d).astype(F)
    t = (v * inv).astype(F)
    r = np.rint(t)
    ref = np.rint((v / d).astype(F))
    outside = _fma(np.abs(t), BAND_REL * np.ones_like(t), np.abs(t - r)) <= _near_limit(np.zeros_like(t))
    assert outside.mean() > 0.45                             # the adversarial half sits inside, the plain half outside
    assert np.array_equal(r[outside], ref[outside])
    # and the band is needed: without it the adversarial inputs do disagree
    assert (r != ref).sum() > 0


def test_rint_div_zp_fast_path_with_the_zero_point_in_the_fma():
    rng = np.random.default_rng(8)
    v, d = _adversarial(rng, 2_000_000, 260)
    z = rng.integers(0, 256, v.shape[0]).astype(F)
    inv = (F(1.0) / d).astype(F)
    t = _fma(v, inv, z)
    r = np.rint(t)
    ref = (np.rint((v / d).astype(F)) + z).astype(F)
    outside = _fma(np.abs(t), BAND_REL * np.ones_like(t), np.abs(t - r)) <= _near_limit(z)
    assert np.array_equal(r[outside], ref[outside])
    # 16-bit codes with a large zero point (sm_abit = 16 quantisers): the relative band still suffices where a fixed 1e-3 did not
    v16, d16 = _adversarial(rng, 1_000_000, 60000)
    z16 = rng.integers(0, 65536, v16.shape[0]).astype(F)
    inv16 = (F(1.0) / d16).astype(F)
    t16 = _fma(v16, inv16, z16)
    r16 = np.rint(t16)
    ref16 = (np.rint((v16 / d16).astype(F)) + z16).astype(F)
    out16 = _fma(np.abs(t16), BAND_REL * np.ones_like(t16), np.abs(t16 - r16)) <= _near_limit(z16)
    assert np.array_equal(r16[out16], ref16[out16])
    fixed = np.abs(t16 - r16) <= F(0.499)                    # rounds 1-3: a fixed band
    assert (r16[fixed] != ref16[fixed]).sum() > 0


def test_softmax_codes_through_one_reciprocal():
    """t = e * fl(1 / fl(sum * delta)) against the reference's fl(fl(e / sum) / delta) (quant_block.py:128-162 on torch's softmax)"""
    rng = np.random.default_rng(9)
    n = 2_000_000
    delta = (F(1.0) / rng.choice([255.0, 65535.0], n)).astype(F)
    s = rng.uniform(1.0, 900.0, n).astype(F)
    k = rng.integers(0, 255, n).astype(np.float64)
    frac = np.where(rng.random(n) < 0.5, 0.5 + rng.integers(-4, 5, n) * 1e-7, rng.random(n))
    e = ((k + frac) * delta.astype(np.float64) * s.astype(np.float64)).astype(F)
    e = np.minimum(e, s)
    inv = (F(1.0) / (s * delta).astype(F)).astype(F)
    t = (e * inv).astype(F)
    r = np.rint(t)
    ref = np.rint(((e / s).astype(F) / delta).astype(F))
    outside = _fma(t, SM_BAND_REL * np.ones_like(t), np.abs(t - r)) <= F(0.5) - F(4e-5)
    assert np.array_equal(r[outside], ref[outside])


def test_the_band_is_rare_on_ordinary_inputs():
    rng = np.random.default_rng(10)
    v = (rng.standard_normal(4_000_000) * 3).astype(F)
    d = F(0.047)
    t = (v * (F(1.0) / d)).astype(F)
    inside = _fma(np.abs(t), BAND_REL * np.ones_like(t), np.abs(t - np.rint(t))) > F(0.5) - F(4e-5)
    p = inside.mean()
    assert p < 3e-4                                          # ~1.2e-4 measured; 2e-3 with the fixed 1e-3 band
    assert 1 - (1 - p) ** 256 < 0.08                         # a wave of 64 lanes x 4 elements takes the exact path in < 8 % of its groups
