// Internal helpers shared by the gfx950 kernels of libedadm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EDADM_EINVAL (-22)
#define EDADM_EIO (-5)
#define EDADM_MAX_DEVICES 64

static inline int edadm_launch_status() { return hipGetLastError() == hipSuccess ? 0 : EDADM_EIO; }

// memory-bound launches: cap the grid at 256 CUs x 8 blocks and grid-stride the rest
static inline int edadm_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}
#define EDADM_RED_BLOCKS 1024  // partial-sum slots used by two-stage reductions

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for 256-thread blocks (4 waves); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* sm4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm4[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = sm4[0] + sm4[1] + sm4[2] + sm4[3];
    __syncthreads();
    return r;
}

// counter RNG: uniform in [0,1) from (seed, index) — splitmix64 finaliser
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// x * sigmoid(x) in the reference's operation order (ddim/models/diffusion.py:27-29)
__device__ __forceinline__ float silu_f(float x) { return x * (1.0f / (1.0f + expf(-x))); }
// the same with the hardware reciprocal (<= 1 ulp) instead of the IEEE division: the fused GroupNorm + SiLU + quantise
// producers of the sampling path are ALU-bound with the division
// and the hardware exponential on x * -log2(e): the library expf spends a dozen instructions on a two-term argument reduction
// that only matters where exp(-x) is either negligible next to 1 or makes the result negligible (|x| > 16); here the
// argument's rounding moves x * sigmoid(x) by at most 1e-8 absolute -- the size of the reciprocal's own last-bit error.
// (The producers are VALU-bound with the library form: GroupNorm + swish -> int8 ran at 4.0-4.6 TB/s against 5.1-5.9 without
// the swish, tools/elem_bw.py.)
// exp(x) for x <= 0 on the hardware exponential (v_exp_f32 of x * log2(e)): the softmax numerators of the sampling path.  Against the
// library routine's two-term argument reduction the argument's rounding adds a relative error of 6e-8 |x| -- for every probability
// that reaches a non-zero 8-bit code (|x| < 6.3) less than the one-ulp error any exponential carries.
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float silu_rcp(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

// erf to < 1 ulp, branch-free: two minimax fits (|x| <= 475/512: odd polynomial; else 1 - exp(-p(|x|))) evaluated
// for every lane and selected -- cheaper on a 64-wide wavefront than the library routine's divergent cases.
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}

// round(v / d) with the reference's true-division result at the cost of a multiply.  t = fl(v * fl(1 / d)) and the reference's
// fl(v / d) are both within 2^-24 |q| (x2 for t: the reciprocal's rounding and the product's) of q = v / d, so the two rounded
// integers can differ only when t lies within 1.8e-7 |t| of a .5 boundary; with the zero point riding in the FMA the sum is
// rounded once more (half an ulp of t) and the reciprocal's error also scales |z|.  Lanes inside the band
//     |t - rint(t)| + 2.4e-7 |t|  >  0.5 - (4e-5 + 1.2e-7 |z|)
// redo the IEEE division behind a REAL branch (the empty asm keeps the compiler from if-converting it into an unconditional
// 10-instruction division per element).  The band is relative: the fixed 1e-3 of rounds 1-3 was 20x wider than needed for 8-bit
// codes -- a wave executes the exact path when ANY of its 64 x NV elements is inside, i.e. 40 % of the 4-element groups instead
// of 3 %, and the quantising GEMM epilogues are bound by exactly this arithmetic -- and too narrow for 16-bit codes above ~5000
// (softmax with sm_abit = 16).  Exact ties are inside the band, so the half-to-even decision is always the reference's.
#define EDADM_BAND_REL 2.4e-7f
__device__ __forceinline__ float near_limit(float z = 0.f) { return 0.5f - fmaf(fabsf(z), 1.2e-7f, 4e-5f); }

__device__ __forceinline__ float rint_div(float v, float d, float inv_d) {
    const float t = v * inv_d;
    float r = rintf(t);
    if (__builtin_expect(fmaf(fabsf(t), EDADM_BAND_REL, fabsf(t - r)) > near_limit(), 0)) {
        asm volatile("" : "+v"(r));
        r = rintf(v / d);
    }
    return r;
}

template <int NV>
__device__ __forceinline__ void rint_div_n(const float (&v)[NV], float d, float inv_d, float (&r)[NV]) {
    float worst = 0.f;                                     // max band distance of the group: one compare, one branch
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float t = v[e] * inv_d;
        r[e] = rintf(t);
        worst = fmaxf(worst, fmaf(fabsf(t), EDADM_BAND_REL, fabsf(t - r[e])));
    }
    if (__builtin_expect(worst > near_limit(), 0)) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            asm volatile("" : "+v"(r[e]));
            r[e] = rintf(v[e] / d);
        }
    }
}

// rint(v / d) + z (z an integer-valued zero point): the addition rides in the multiply (one FMA)
template <int NV>
__device__ __forceinline__ void rint_div_zp_n(const float (&v)[NV], float d, float inv_d, float z, float (&r)[NV]) {
    float worst = 0.f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float t = fmaf(v[e], inv_d, z);
        r[e] = rintf(t);
        worst = fmaxf(worst, fmaf(fabsf(t), EDADM_BAND_REL, fabsf(t - r[e])));
    }
    if (__builtin_expect(worst > near_limit(z), 0)) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            asm volatile("" : "+v"(r[e]));
            r[e] = rintf(v[e] / d) + z;
        }
    }
}

// GEGLU + quantise for the GEMM epilogue: codes r[e] = rint((val * gelu(gate)) / d) + z for NV (value, gate) pairs, equal
// BIT FOR BIT to the exact form (erf_fast to < 1 ulp, IEEE division) at a third of its instruction count -- this epilogue is bound
// by the vector ALU (tools/stamps.py: 12 k cycles per 256 x 192 tile against 3.7 k of MFMA), so the fast pass is written for
// instruction count: 17 per pair.
// Fast pass: with pe = poly(u) u exp(-g^2 / 2), u = 1 / (1 + p |g| / sqrt 2) (Abramowitz-Stegun 7.1.26: erf(|g| / sqrt 2) = 1 - pe,
// |error| <= 1.5e-7), gate * (1 + erf(gate / sqrt 2)) = g + |g| (1 - pe) for either sign of g -- no select, no separate 2 - pe -- so
//     t = val * (g + |g| (1 - pe)) * (0.5 / d) + z.
// The absolute error of g + |g| (1 - pe) is |g| x (1.5e-7 + the roundings of pe, of 1 - pe and of the sum: 5e-7 in all), that of the
// exact form's y a few 2^-24 |y|: the rounded code can differ from the exact form's only if t lies closer to a .5 boundary than
// |val gate| / d * 1.2e-6 + 6e-5 (roundings of t itself and of the exact form's quotient, |t| < 512; beyond that the clamp
// saturates).  Groups with such an element redo all NV the exact way behind a real branch (a few % of the wave groups).
template <int NV>
__device__ __forceinline__ void geglu_codes_n(const float (&val)[NV], const float (&gate)[NV], float d, float inv_d, float z,
                                              float (&r)[NV]) {
    float worst = 0.f;
    const float hd = 0.5f * inv_d, kb = inv_d * 1.2e-6f;     // exact scaling / bound coefficient: uniform, hoisted
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float g = gate[e];
        // x = |g| / sqrt 2 never materialises: 1 + p x = fma(|g|, p / sqrt 2, 1), and exp(-x^2) = exp2(-(g sqrt(log2(e) / 2))^2)
        const float u = __builtin_amdgcn_rcpf(fmaf(fabsf(g), 0.3275911f * 0.70710678118654752440f, 1.0f));
        const float xe = g * 0.84932180028801904272f;
        float pl = fmaf(1.061405429f, u, -1.453152027f);
        pl = fmaf(pl, u, 1.421413741f);
        pl = fmaf(pl, u, -0.284496736f);
        pl = fmaf(pl, u, 0.254829592f);
        const float om = fmaf(-__builtin_amdgcn_exp2f(-(xe * xe)), pl * u, 1.0f);      // erf(|g| / sqrt 2)
        const float h = fmaf(fabsf(g), om, g);                                          // g (1 + erf(g / sqrt 2))
        const float t = fmaf(val[e] * h, hd, z);
        r[e] = rintf(t);
        worst = fmaxf(worst, fmaf(fabsf(val[e]) * fabsf(g), kb, fabsf(t - r[e])));
    }
    if (__builtin_expect(worst > 0.5f - 6e-5f, 0)) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            asm volatile("" : "+v"(r[e]));
            const float y = val[e] * (0.5f * gate[e] * (1.0f + erf_fast(gate[e] * 0.70710678118654752440f)));
            r[e] = rintf(y / d) + z;
        }
    }
}

// clamp to [lo, hi] in one instruction (v_med3_f32)
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// four quantised codes (float, already clamped to [0, 255]) -> int8 operand bytes code - 128
__device__ __forceinline__ uint32_t pack_codes_i8(const float (&q)[4]) {
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(q[0], 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(q[1], 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(q[2], 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(q[3], 3, w);
    return w ^ 0x80808080u;
}
