// K4w -- int8-activation x INT4-WEIGHT GEMM that reads the weights as packed nibbles (two 4-bit codes per byte, the storage
// format of edadm/state.py / edadm_pack_w4) and expands them in registers on their way into the int8 MFMA: for the few-row layers
// (time-embedding tables of a sampling run, the one-token cross-attention branches, any dense layer at M <= 2048) the weights ARE
// the bytes of the launch, and at 4 bits they are half of what the int8 copy costs (quant_layer.py:406-437 at inference; north_star:
// "per-channel int4 weight ... fused into the GEMMs").
//
//   out[m][n] = scale[n] * sum_k a[m][k] * (code[n][k] - zp4[n]) + bias[n] (+ rowadd[m / rpb][n]) (+ residual[m][n])
//
// The nibble codes (0..15) go into the MFMA as they are; the per-row zero point comes off afterwards through the row sums of A,
// which a second MFMA against an all-ones fragment accumulates alongside (sum_k a (c - z) = sum_k a c - z sum_k a): integer
// arithmetic throughout, so the result has the bits of the int8 kernels on the unpacked weights.
#include "common.h"
#include "../../include/edadm.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 64 x 64 output tile, 4 waves of 32 x 32; K in steps of 32 (one v_mfma_i32_32x32x32_i8 per wave and step + one for the row sums)
__global__ void __launch_bounds__(256) k_gemm_w4(const int8_t* __restrict__ A, int64_t lda, const uint8_t* __restrict__ W4,
                                                 const float* __restrict__ zp4, int64_t M, int64_t N, int64_t K,
                                                 const float* __restrict__ scale, const float* __restrict__ bias,
                                                 const float* __restrict__ rowadd, int64_t rpb, const float* __restrict__ residual,
                                                 int64_t ldr, float* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * 64 + (wave >> 1) * 32, n0 = (int64_t)blockIdx.x * 64 + (wave & 1) * 32;
    const int64_t am = m0 + fr, wn = n0 + fr;
    const bool a_ok = am < M, w_ok = wn < N;
    const int8_t* ap = A + (a_ok ? am : 0) * lda + fh * 16;
    const uint8_t* wp = W4 + (w_ok ? wn : 0) * (K >> 1) + fh * 8;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, rs = acc;
    const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const v4i zero4 = {0, 0, 0, 0};
    auto expand = [&](const uint2 pk, v4i& fb) {
        const uint32_t l0 = pk.x & 0x0f0f0f0fu, h0 = (pk.x >> 4) & 0x0f0f0f0fu;          // even / odd elements of bytes 0..3
        const uint32_t l1 = pk.y & 0x0f0f0f0fu, h1 = (pk.y >> 4) & 0x0f0f0f0fu;
        fb[0] = (int)__builtin_amdgcn_perm(h0, l0, 0x05010400u);                         // (l0 h0 l1 h1) of the low bytes
        fb[1] = (int)__builtin_amdgcn_perm(h0, l0, 0x07030602u);
        fb[2] = (int)__builtin_amdgcn_perm(h1, l1, 0x05010400u);
        fb[3] = (int)__builtin_amdgcn_perm(h1, l1, 0x07030602u);
    };
    // four K-steps per trip, all eight loads issued before the first MFMA: at M of a few rows the loop is a latency chain
    // (one L2 round trip per step otherwise: 26 us for 1280 x 1280 at 8 rows)
    int64_t k0 = 0;
    for (; k0 + 128 <= K; k0 += 128) {
        v4i fa[4];
        uint2 pk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            fa[u] = a_ok ? *reinterpret_cast<const v4i*>(ap + k0 + 32 * u) : zero4;
            pk[u] = w_ok ? *reinterpret_cast<const uint2*>(wp + ((k0 + 32 * u) >> 1)) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v4i fb;
            expand(pk[u], fb);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u], fb, acc, 0, 0, 0);
            rs = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u], ones, rs, 0, 0, 0);
        }
    }
    for (; k0 < K; k0 += 32) {
        v4i fa = a_ok ? *reinterpret_cast<const v4i*>(ap + k0) : zero4, fb;
        expand(w_ok ? *reinterpret_cast<const uint2*>(wp + (k0 >> 1)) : make_uint2(0u, 0u), fb);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
        rs = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, ones, rs, 0, 0, 0);
    }
    if (!w_ok) return;
    const int z = (int)zp4[wn];
    const float s = scale[wn], bs = bias ? bias[wn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + 8 * (r >> 2) + 4 * fh + (r & 3);
        if (row >= M) continue;
        float v = fmaf((float)(acc[r] - z * rs[r]), s, bs);          // the int8 kernels' epilogue, operation by operation
        if (rowadd) v += rowadd[(row / rpb) * N + wn];
        if (residual) v += residual[row * ldr + wn];
        out[row * ldo + wn] = v;
    }
}

// M <= 64: the same arithmetic with the four waves of a workgroup on ONE 32 x 32 output tile, each taking every fourth K-step, the
// partial accumulators (and row sums) added up through LDS: N / 32 workgroups x 4 waves keep 4x the loads of the tile form in flight
// -- at a handful of rows the launch is a latency chain over the weight stream (1280 x 1280 at 8 rows: 26 us in the tile form)
__global__ void __launch_bounds__(256) k_gemm_w4_splitk(const int8_t* __restrict__ A, int64_t lda, const uint8_t* __restrict__ W4,
                                                        const float* __restrict__ zp4, int64_t M, int64_t N, int64_t K,
                                                        const float* __restrict__ scale, const float* __restrict__ bias,
                                                        const float* __restrict__ rowadd, int64_t rpb,
                                                        const float* __restrict__ residual, int64_t ldr, float* __restrict__ out,
                                                        int64_t ldo) {
    __shared__ int red[3][2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
    const int64_t am = m0 + fr, wn = n0 + fr;
    const bool a_ok = am < M, w_ok = wn < N;
    const int8_t* ap = A + (a_ok ? am : 0) * lda + fh * 16;
    const uint8_t* wp = W4 + (w_ok ? wn : 0) * (K >> 1) + fh * 8;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, rs = acc;
    const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const v4i zero4 = {0, 0, 0, 0};
    auto expand = [&](const uint2 pk, v4i& fb) {
        const uint32_t l0 = pk.x & 0x0f0f0f0fu, h0 = (pk.x >> 4) & 0x0f0f0f0fu;
        const uint32_t l1 = pk.y & 0x0f0f0f0fu, h1 = (pk.y >> 4) & 0x0f0f0f0fu;
        fb[0] = (int)__builtin_amdgcn_perm(h0, l0, 0x05010400u);
        fb[1] = (int)__builtin_amdgcn_perm(h0, l0, 0x07030602u);
        fb[2] = (int)__builtin_amdgcn_perm(h1, l1, 0x05010400u);
        fb[3] = (int)__builtin_amdgcn_perm(h1, l1, 0x07030602u);
    };
    const int64_t nsteps = K >> 5;
    int64_t st = wave;                                             // this wave's K-steps: wave, wave + 4, ...
    for (; st + 12 < nsteps; st += 16) {
        v4i fa[4];
        uint2 pk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k0 = (st + 4 * u) << 5;
            fa[u] = a_ok ? *reinterpret_cast<const v4i*>(ap + k0) : zero4;
            pk[u] = w_ok ? *reinterpret_cast<const uint2*>(wp + (k0 >> 1)) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v4i fb;
            expand(pk[u], fb);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u], fb, acc, 0, 0, 0);
            rs = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u], ones, rs, 0, 0, 0);
        }
    }
    for (; st < nsteps; st += 4) {
        const int64_t k0 = st << 5;
        v4i fa = a_ok ? *reinterpret_cast<const v4i*>(ap + k0) : zero4, fb;
        expand(w_ok ? *reinterpret_cast<const uint2*>(wp + (k0 >> 1)) : make_uint2(0u, 0u), fb);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
        rs = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, ones, rs, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            red[wave - 1][0][r][lane] = acc[r];
            red[wave - 1][1][r][lane] = rs[r];
        }
    }
    __syncthreads();
    if (wave > 0 || !w_ok) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] += red[w][0][r][lane];
            rs[r] += red[w][1][r][lane];
        }
    const int z = (int)zp4[wn];
    const float s = scale[wn], bs = bias ? bias[wn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + 8 * (r >> 2) + 4 * fh + (r & 3);
        if (row >= M) continue;
        float v = fmaf((float)(acc[r] - z * rs[r]), s, bs);
        if (rowadd) v += rowadd[(row / rpb) * N + wn];
        if (residual) v += residual[row * ldr + wn];
        out[row * ldo + wn] = v;
    }
}

extern "C" int edadm_qgemm_w4(const int8_t* A, int64_t lda, const uint8_t* W4, const float* zp4, int64_t M, int64_t N, int64_t K,
                              const float* scale, const float* bias, const float* rowadd, int64_t rows_per_batch,
                              const float* residual, int64_t ldr, float* out, int64_t ldo, void* stream) {
    if (!A || !W4 || !zp4 || !scale || !out || M <= 0 || N <= 0 || K <= 0 || (K & 31) || (lda & 15) || lda < K || ldo < N)
        return EDADM_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)W4 & 7)) return EDADM_EINVAL;
    if (rowadd && rows_per_batch <= 0) return EDADM_EINVAL;
    if (residual && ldr < N) return EDADM_EINVAL;
    if (M <= 64)
        hipLaunchKernelGGL(k_gemm_w4_splitk, dim3((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                           A, lda, W4, zp4, M, N, K, scale, bias, rowadd, rowadd ? rows_per_batch : 1, residual, ldr, out, ldo);
    else
        hipLaunchKernelGGL(k_gemm_w4, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, A,
                           lda, W4, zp4, M, N, K, scale, bias, rowadd, rowadd ? rows_per_batch : 1, residual, ldr, out, ldo);
    return edadm_launch_status();
}
