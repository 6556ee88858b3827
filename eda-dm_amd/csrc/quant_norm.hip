// Producers of MFMA operands for the quantised sampling path: activation quantisation (inference
// form of K1), fused GroupNorm / LayerNorm / SiLU / GEGLU + quantise (K5), softmax + quantise
// (K6), layout changes.  All HBM-bound: each input element is read once (GroupNorm: twice — one
// statistics pass, one apply pass) and each operand byte written once; 16-byte accesses.
#include "common.h"
#include <stdlib.h>
#include "../../include/edadm.h"
#include <hip/hip_fp16.h>

// (delta, zero point, qmax) as the host passes them; `inv` (the 4th float, unused by the host) is filled with
// 1/delta when a kernel loads the entry: codes are rint(x * inv) with the exact division kept for the rare values
// inside the rounding-boundary band (rint_div, common.h) -- the IEEE division costs ~12 instructions per element
// and made these producers ALU-bound instead of HBM-bound.
struct QP { float d, z, qmax, inv; };
__device__ __forceinline__ QP qp_load(const QP* p, int i) {
    QP q = p[i];
    q.inv = 1.0f / q.d;
    return q;
}

__device__ __forceinline__ float q_code_f(float x, const QP& q) {      // clamped code as a float
    return fminf(fmaxf(rint_div(x, q.d, q.inv) + q.z, 0.f), q.qmax);
}
__device__ __forceinline__ int q_code_i8(float x, const QP& q) { return (int)q_code_f(x, q) - 128; }
__device__ __forceinline__ uint32_t pack4_i8(int a, int b, int c, int d) {
    return (uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8) | ((uint32_t)(c & 0xff) << 16) |
           ((uint32_t)(d & 0xff) << 24);
}
__device__ __forceinline__ uint32_t quant4(const float4& v, const QP& q) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    float r[4];
    rint_div_zp_n<4>(x, q.d, q.inv, q.z, r);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = clampf(r[e], 0.f, q.qmax);
    return pack_codes_i8(r);
}

// ------------------------------------------------------------------ plain quantise [rows][C]
// x is either one [rows][C] matrix or the channel concatenation [x | x2] of two (C1 and C - C1 channels wide):
// the UNet skip concatenation is never materialised in fp32, its consumers read the two halves in place.
__global__ void __launch_bounds__(256) k_quant_i8(const float* __restrict__ x, const float* __restrict__ x2, int64_t C1,
                                                  int8_t* __restrict__ out, int64_t rows, int64_t C,
                                                  const QP* __restrict__ qp, int64_t split, int64_t rows2) {
    // rows2 > 0: x2 holds rows2 < rows rows and is read periodically (r % rows2): the half-batch skip tensors of a
    // classifier-free-guidance pair, whose two halves are identical up to the first context-dependent layer
    const int64_t n = rows * C, stride = (int64_t)gridDim.x * blockDim.x;
    const QP q0 = qp_load(qp, 0);
    const QP q1 = split > 0 ? qp_load(qp, 1) : q0;
    if ((C & 3) == 0 && (split & 3) == 0 && (C1 & 3) == 0) {
        const int64_t n4 = n >> 2, C4 = C >> 2, s4 = split >> 2, A4 = C1 >> 2, B4 = C4 - A4;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const int64_t r = i / C4, c = i - r * C4;
            const float4 v = c < A4 ? reinterpret_cast<const float4*>(x)[r * A4 + c]
                                    : reinterpret_cast<const float4*>(x2)[(rows2 > 0 ? r % rows2 : r) * B4 + (c - A4)];
            const bool second = split > 0 && c >= s4;
            reinterpret_cast<uint32_t*>(out)[i] = quant4(v, second ? q1 : q0);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const int64_t r = i / C, c = i - r * C;
            const float v = c < C1 ? x[r * C1 + c] : x2[(rows2 > 0 ? r % rows2 : r) * (C - C1) + (c - C1)];
            const bool second = split > 0 && c >= split;
            out[i] = (int8_t)q_code_i8(v, second ? q1 : q0);
        }
    }
}
extern "C" int edadm_quant_i8_cat_rep(const float* x1, int64_t C1, const float* x2, int64_t C2, int8_t* out, int64_t rows,
                                      const float* qp, int64_t split, int64_t rows2, void* stream) {
    const int64_t C = C1 + (x2 ? C2 : 0);
    if (!x1 || !out || !qp || rows <= 0 || C1 <= 0 || (x2 && C2 <= 0) || split < 0 || split >= C + (split == 0) ||
        rows2 < 0 || (rows2 > 0 && (!x2 || rows % rows2)))
        return EDADM_EINVAL;
    hipLaunchKernelGGL(k_quant_i8, dim3(edadm_grid(rows * C / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x1, x2,
                       C1, out, rows, C, (const QP*)qp, split, rows2);
    return edadm_launch_status();
}
extern "C" int edadm_quant_i8_cat(const float* x1, int64_t C1, const float* x2, int64_t C2, int8_t* out, int64_t rows,
                                  const float* qp, int64_t split, void* stream) {
    return edadm_quant_i8_cat_rep(x1, C1, x2, C2, out, rows, qp, split, 0, stream);
}
extern "C" int edadm_quant_i8(const float* x, int8_t* out, int64_t rows, int64_t C, const float* qp, int64_t split,
                              void* stream) {
    return edadm_quant_i8_cat(x, C, nullptr, 0, out, rows, qp, split, stream);
}

// f16 operand = code - zp (exact integers), optional pre-multiplier (q*scale, openaimodel.py:391)
__global__ void __launch_bounds__(256) k_quant_f16(const float* __restrict__ x, int64_t ldx,
                                                   __half* __restrict__ out, int64_t ldo, int64_t rows, int64_t C,
                                                   const QP* __restrict__ qp, float premul) {
    const QP q = qp_load(qp, 0);
    const int64_t n = rows * C, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / C, c = i - r * C;
        float v = x[r * ldx + c];
        if (premul != 1.0f) v = v * premul;
        const float code = q_code_f(v, q);
        out[r * ldo + c] = __float2half(code - q.z);
    }
}
extern "C" int edadm_quant_f16(const float* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int64_t C,
                               const float* qp, float premul, void* stream) {
    if (!x || !out || !qp || rows <= 0 || C <= 0 || ldx < C || ldo < C) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_quant_f16, dim3(edadm_grid(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (__half*)out, ldo, rows, C, (const QP*)qp, premul);
    return edadm_launch_status();
}

// the legacy AttentionBlock's qkv tensor [rows][heads x (q | k | v) x d] (openaimodel.py:390-393) in ONE pass: column c belongs to
// group (c / d) % 3, each group with its own quantiser (qp[0..2]) and pre-multiplier; same layout out, f16 codes minus zero point
__global__ void __launch_bounds__(256) k_quant_f16_qkv(const float* __restrict__ x, int64_t ldx, __half* __restrict__ out,
                                                       int64_t ldo, int64_t rows, int64_t C, int64_t d,
                                                       const QP* __restrict__ qp, float pm0, float pm1, float pm2) {
    const QP q0 = qp_load(qp, 0), q1 = qp_load(qp, 1), q2 = qp_load(qp, 2);
    const int64_t C4 = C >> 2, n = rows * C4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / C4, c = (i - r * C4) * 4;
        const int grp = (int)((c / d) % 3);
        const QP& q = grp == 0 ? q0 : grp == 1 ? q1 : q2;
        const float pm = grp == 0 ? pm0 : grp == 1 ? pm1 : pm2;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float e[4] = {v.x, v.y, v.z, v.w};
        __half h[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = __float2half(q_code_f(pm != 1.0f ? e[k] * pm : e[k], q) - q.z);
        *reinterpret_cast<uint2*>(out + r * ldo + c) = *reinterpret_cast<const uint2*>(h);
    }
}
extern "C" int edadm_quant_f16_qkv(const float* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int64_t C, int64_t d,
                                   const float* qp3, float premul_q, float premul_k, float premul_v, void* stream) {
    if (!x || !out || !qp3 || rows <= 0 || C <= 0 || d <= 0 || (d & 3) || C % (3 * d) || ldx < C || ldo < C || (ldx & 3) ||
        (ldo & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 7))
        return EDADM_EINVAL;
    hipLaunchKernelGGL(k_quant_f16_qkv, dim3(edadm_grid(rows * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (__half*)out, ldo, rows, C, d, (const QP*)qp3, premul_q, premul_k, premul_v);
    return edadm_launch_status();
}

// ------------------------------------------------------------------ NCHW <-> NHWC (model boundary)
__global__ void __launch_bounds__(256) k_nchw_to_nhwc(const float* __restrict__ x, float* __restrict__ out,
                                                      int64_t B, int64_t C, int64_t HW) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.z;
    const int64_t hw0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int64_t c = c0 + j, hw = hw0 + tx;
        tile[j][tx] = (c < C && hw < HW) ? x[(b * C + c) * HW + hw] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t hw = hw0 + j, c = c0 + tx;
        if (c < C && hw < HW) out[(b * HW + hw) * C + c] = tile[tx][j];
    }
}
extern "C" int edadm_nchw_to_nhwc(const float* x, float* out, int64_t B, int64_t C, int64_t HW, void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || HW <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B),
                       dim3(256), 0, (hipStream_t)stream, x, out, B, C, HW);
    return edadm_launch_status();
}
extern "C" int edadm_nhwc_to_nchw(const float* x, float* out, int64_t B, int64_t C, int64_t HW, void* stream) {
    // the same transpose with the roles of C and HW exchanged
    if (!x || !out || B <= 0 || C <= 0 || HW <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((C + 31) / 32), (unsigned)((HW + 31) / 32), (unsigned)B),
                       dim3(256), 0, (hipStream_t)stream, x, out, B, HW, C);
    return edadm_launch_status();
}

// im2col + quantise for tiny-Cin 3x3/pad-1 convolutions: out[m][Kpad], k = (ky*3+kx)*C + c
__global__ void __launch_bounds__(256) k_im2col_q(const float* __restrict__ x, int8_t* __restrict__ out,
                                                  int64_t B, int64_t H, int64_t W, int64_t C, int64_t Kpad,
                                                  const QP* __restrict__ qp) {
    const QP q = qp_load(qp, 0);
    const int padv = (int)q.z - 128;
    const int64_t n = B * H * W * Kpad, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t m = i / Kpad, k = i - m * Kpad;
        int v = 0;
        if (k < 9 * C) {
            const int64_t tap = k / C, c = k - tap * C;
            const int64_t b = m / (H * W), r = m - b * H * W, y = r / W, xx = r - y * W;
            const int64_t iy = y + tap / 3 - 1, ix = xx + tap % 3 - 1;
            v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? q_code_i8(x[((b * H + iy) * W + ix) * C + c], q) : padv;
        }
        out[i] = (int8_t)v;
    }
}
extern "C" int edadm_im2col_quant_i8(const float* x, int8_t* out, int64_t B, int64_t H, int64_t W, int64_t C,
                                     int64_t Kpad, const float* qp, void* stream) {
    if (!x || !out || !qp || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Kpad < 9 * C) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_im2col_q, dim3(edadm_grid(B * H * W * Kpad, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       out, B, H, W, C, Kpad, (const QP*)qp);
    return edadm_launch_status();
}

// ------------------------------------------------------------------ GroupNorm on NHWC
// pass 1: per (b, chunk of rows) block, per-channel (sum, sumsq) partials -> ws[b][chunk][C][2];
// finalize: per (b, group) sum of chunks x channels-in-group in double -> stats[b][g] = {mean, rstd}
#define GN_CHUNKS(HW) ((int)((HW) >= 1024 ? 32 : ((HW) >= 64 ? 8 : 1)))

__global__ void __launch_bounds__(256) k_gn_partial(const float* __restrict__ x, const float* __restrict__ x2, int64_t C1,
                                                    float* __restrict__ ws, int64_t HW, int64_t C, int nchunk, int64_t B2) {
    extern __shared__ float sm[];  // [RS][C][2] when RS > 1
    const int64_t b = blockIdx.y;
    const int64_t b2 = B2 > 0 ? b % B2 : b;                          // x2 of B2 images read periodically (CFG pair)
    const int chunk = blockIdx.x;
    const int64_t r0 = HW * chunk / nchunk, r1 = HW * (chunk + 1) / nchunk;
    const int Q = (int)(C >> 2);
    const int RS = Q <= 256 ? 256 / Q : 1;
    const int tid = threadIdx.x;
    const int Q1 = (int)(C1 >> 2), Q2 = Q - Q1;                      // [x | x2] halves (x2 == nullptr: Q1 == Q)
    const float4* xa = reinterpret_cast<const float4*>(x) + b * HW * Q1;
    const float4* xc = reinterpret_cast<const float4*>(x2) + b2 * HW * Q2;
    float* wb = ws + ((b * nchunk + chunk) * C) * 2;
    if (Q <= 256) {
        const int q = tid % Q, rs = tid / Q;
        float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        if (rs < RS) {
            const bool first = q < Q1;
            // four 16-byte loads in flight per lane (one at a time the pass waited out every load: 92 % of a wave's life parked,
            // rocprofv3 --pmc SQ_WAIT_ANY); the sums still run over the rows in their order: the same bits
            const float4* src = first ? xa + q : xc + (q - Q1);
            const int64_t st = first ? Q1 : Q2;
            auto acc = [&](const float4 v) {
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                ss[0] += v.x * v.x; ss[1] += v.y * v.y; ss[2] += v.z * v.z; ss[3] += v.w * v.w;
            };
            int64_t r = r0 + rs;
            for (; r + 3 * RS < r1; r += 4 * RS) {
                const float4 v0 = src[r * st], v1 = src[(r + RS) * st], v2 = src[(r + 2 * RS) * st], v3 = src[(r + 3 * RS) * st];
                acc(v0); acc(v1); acc(v2); acc(v3);
            }
            for (; r < r1; r += RS) acc(src[r * st]);
            for (int j = 0; j < 4; ++j) {
                sm[((int64_t)rs * C + q * 4 + j) * 2] = s[j];
                sm[((int64_t)rs * C + q * 4 + j) * 2 + 1] = ss[j];
            }
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float a = 0.f, bq = 0.f;
            for (int r = 0; r < RS; ++r) { a += sm[((int64_t)r * C + c) * 2]; bq += sm[((int64_t)r * C + c) * 2 + 1]; }
            wb[2 * c] = a;
            wb[2 * c + 1] = bq;
        }
    } else {
        for (int q = tid; q < Q; q += 256) {
            float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
            const bool first = q < Q1;
            for (int64_t r = r0; r < r1; ++r) {
                const float4 v = first ? xa[r * Q1 + q] : xc[r * Q2 + (q - Q1)];
                s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
                ss[0] += v.x * v.x; ss[1] += v.y * v.y; ss[2] += v.z * v.z; ss[3] += v.w * v.w;
            }
            for (int j = 0; j < 4; ++j) { wb[2 * (q * 4 + j)] = s[j]; wb[2 * (q * 4 + j) + 1] = ss[j]; }
        }
    }
}
// ws2 != nullptr: channels [C1, C) come from a second partials buffer (the other half of a skip concatenation)
__global__ void __launch_bounds__(64) k_gn_final(const float* __restrict__ ws, const float* __restrict__ ws2, int64_t C1,
                                                 float* __restrict__ stats,
                                                 int64_t HW, int64_t C, int64_t G, int nchunk, float eps, int64_t B2 = 0,
                                                 int nchunk2 = 0) {
    // nchunk2: slabs per image of the second buffer when its producer cut the image differently (0: as the first)
    const int64_t b = blockIdx.y, g = blockIdx.x;
    const int64_t b2 = B2 > 0 ? b % B2 : b;               // the second buffer may hold a whole fraction of the batch (guidance pair)
    const int cpg = (int)(C / G);
    const int nc2 = nchunk2 > 0 ? nchunk2 : nchunk;
    double s = 0.0, ss = 0.0;
    const int items = (nchunk > nc2 ? nchunk : nc2) * cpg;
    for (int i = threadIdx.x; i < items; i += 64) {
        const int ch = i / cpg, c = (int)(g * cpg) + i % cpg;
        const bool first = !ws2 || c < C1;
        if (ch >= (first ? nchunk : nc2)) continue;
        const float* p = first ? ws + ((b * nchunk + ch) * (ws2 ? C1 : C) + c) * 2
                               : ws2 + ((b2 * nc2 + ch) * (C - C1) + (c - C1)) * 2;
        s += (double)p[0];
        ss += (double)p[1];
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    if (threadIdx.x == 0) {
        const double n = (double)HW * cpg;
        const double mean = s / n;
        double var = ss / n - mean * mean;
        if (var < 0) var = 0;
        stats[(b * G + g) * 2] = (float)mean;
        stats[(b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
extern "C" int64_t edadm_gn_ws_floats(int64_t B, int64_t HW, int64_t C) { return B * GN_CHUNKS(HW) * C * 2; }
extern "C" int edadm_groupnorm_stats_cat_rep(const float* x1, int64_t C1, const float* x2, int64_t C2, float* stats, float* ws,
                                             int64_t B, int64_t HW, int64_t G, float eps, int64_t B2, void* stream) {
    const int64_t C = C1 + (x2 ? C2 : 0);
    if (!x1 || !stats || !ws || B <= 0 || HW <= 0 || C1 <= 0 || (x2 && C2 <= 0) || G <= 0 || (C % G) || (C & 3) || (C1 & 3) ||
        B2 < 0 || (B2 > 0 && (!x2 || B % B2)))
        return EDADM_EINVAL;
    const int nchunk = GN_CHUNKS(HW);
    const int Q = (int)(C >> 2);
    const int RS = Q <= 256 ? 256 / Q : 1;
    const size_t smem = Q <= 256 ? (size_t)RS * C * 2 * sizeof(float) : 0;
    hipLaunchKernelGGL(k_gn_partial, dim3(nchunk, (unsigned)B), dim3(256), smem, (hipStream_t)stream, x1, x2, C1, ws, HW,
                       C, nchunk, B2);
    hipLaunchKernelGGL(k_gn_final, dim3((unsigned)G, (unsigned)B), dim3(64), 0, (hipStream_t)stream, ws, (const float*)nullptr,
                       C, stats, HW, C, G, nchunk, eps);
    return edadm_launch_status();
}
extern "C" int edadm_groupnorm_stats_cat(const float* x1, int64_t C1, const float* x2, int64_t C2, float* stats, float* ws,
                                         int64_t B, int64_t HW, int64_t G, float eps, void* stream) {
    return edadm_groupnorm_stats_cat_rep(x1, C1, x2, C2, stats, ws, B, HW, G, eps, 0, stream);
}
// pass 2 alone, over per-channel partials [B][nchunk][C][2] that a producer already wrote (edadm_qgemm_i8_gn)
extern "C" int edadm_groupnorm_final_cat_rep(const float* ws1, int64_t C1, const float* ws2, int64_t C2, float* stats, int64_t B,
                                             int64_t HW, int64_t G, int64_t nchunk, float eps, int64_t B2, void* stream) {
    const int64_t C = C1 + (ws2 ? C2 : 0);
    if (!ws1 || !stats || B <= 0 || HW <= 0 || C1 <= 0 || (ws2 && C2 <= 0) || G <= 0 || (C % G) || nchunk <= 0 || B2 < 0 ||
        (B2 > 0 && (!ws2 || B % B2)))
        return EDADM_EINVAL;
    hipLaunchKernelGGL(k_gn_final, dim3((unsigned)G, (unsigned)B), dim3(64), 0, (hipStream_t)stream, ws1, ws2, C1, stats, HW, C,
                       G, (int)nchunk, eps, B2);
    return edadm_launch_status();
}
// the same with its own slab count for the second buffer (producers with 64-row and 32-row slabs on the two halves)
extern "C" int edadm_groupnorm_final_cat_rep2(const float* ws1, int64_t C1, const float* ws2, int64_t C2, float* stats, int64_t B,
                                              int64_t HW, int64_t G, int64_t nchunk1, int64_t nchunk2, float eps, int64_t B2,
                                              void* stream) {
    const int64_t C = C1 + (ws2 ? C2 : 0);
    if (!ws1 || !stats || B <= 0 || HW <= 0 || C1 <= 0 || (ws2 && C2 <= 0) || G <= 0 || (C % G) || nchunk1 <= 0 || nchunk2 < 0 ||
        B2 < 0 || (B2 > 0 && (!ws2 || B % B2)))
        return EDADM_EINVAL;
    hipLaunchKernelGGL(k_gn_final, dim3((unsigned)G, (unsigned)B), dim3(64), 0, (hipStream_t)stream, ws1, ws2, C1, stats, HW, C,
                       G, (int)nchunk1, eps, B2, (int)nchunk2);
    return edadm_launch_status();
}
extern "C" int edadm_groupnorm_final_cat(const float* ws1, int64_t C1, const float* ws2, int64_t C2, float* stats, int64_t B,
                                         int64_t HW, int64_t G, int64_t nchunk, float eps, void* stream) {
    return edadm_groupnorm_final_cat_rep(ws1, C1, ws2, C2, stats, B, HW, G, nchunk, eps, 0, stream);
}
extern "C" int edadm_groupnorm_stats(const float* x, float* stats, float* ws, int64_t B, int64_t HW, int64_t C,
                                     int64_t G, float eps, void* stream) {
    return edadm_groupnorm_stats_cat(x, C, nullptr, 0, stats, ws, B, HW, G, eps, stream);
}
__global__ void __launch_bounds__(256) k_gn_apply(const float* __restrict__ x, const float* __restrict__ x2, int64_t C1,
                                                  const float* __restrict__ stats,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* __restrict__ scale_shift, int64_t HW, int64_t C,
                                                  int64_t G, int silu, float* __restrict__ out_f32,
                                                  int8_t* __restrict__ q0, int8_t* __restrict__ q1,
                                                  int8_t* __restrict__ q2, const QP* __restrict__ qp, int nq,
                                                  int rows_per_block, int8_t* __restrict__ qraw,
                                                  const QP* __restrict__ qpr, int64_t raw_split, int64_t B2) {
    const int64_t b = blockIdx.y;
    const int64_t b2 = B2 > 0 ? b % B2 : b;                          // x2 of B2 images read periodically (CFG pair)
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < HW ? r0 + rows_per_block : HW;
    const int Q = (int)(C >> 2);
    const int cpg = (int)(C / G);
    QP qa = qp ? qp_load(qp, 0) : QP{1, 0, 255, 1}, qb = (qp && nq > 1) ? qp_load(qp, 1) : qa, qc = (qp && nq > 2) ? qp_load(qp, 2) : qa;
    const int RS = Q <= 256 ? 256 / Q : 1;
    const int tid = threadIdx.x;
    const int qstep = Q <= 256 ? Q : 256;           // a thread keeps its quad(s): constants loaded once
    const int rs = Q <= 256 ? tid / Q : 0;
    if (Q <= 256 && rs >= RS) return;
    for (int q = Q <= 256 ? tid % Q : tid; q < Q; q += qstep) {
        float a[4], bb[4], sc[4], sh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = q * 4 + j;
            const int g = c / cpg;
            const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
            a[j] = rstd * gamma[c];
            bb[j] = beta[c] - mean * a[j];
            sc[j] = 1.f; sh[j] = 0.f;
            if (scale_shift) { sc[j] = 1.0f + scale_shift[b * 2 * C + c]; sh[j] = scale_shift[b * 2 * C + C + c]; }
        }
        const int Q1 = (int)(C1 >> 2), Q2 = Q - Q1;
        const bool first = q < Q1;
        // the un-normalised input quantised for a second consumer of the same tensor (the ResBlock's skip convolution,
        // two quantisers over channel ranges when raw_split > 0): what edadm_quant_i8_cat computes, without its read
        QP qr = qa;
        if (qraw) qr = qp_load(qpr, (raw_split > 0 && q * 4 >= raw_split) ? 1 : 0);
        for (int64_t r = r0 + rs; r < r1; r += RS) {
            const int64_t idx = (b * HW + r) * Q + q;
            float4 v = first ? reinterpret_cast<const float4*>(x)[(b * HW + r) * Q1 + q]
                             : reinterpret_cast<const float4*>(x2)[(b2 * HW + r) * Q2 + (q - Q1)];
            float y[4] = {v.x * a[0] + bb[0], v.y * a[1] + bb[1], v.z * a[2] + bb[2], v.w * a[3] + bb[3]};
            if (scale_shift) {
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = y[j] * sc[j] + sh[j];
            }
            if (silu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = silu_rcp(y[j]);
            }
            const float4 o = make_float4(y[0], y[1], y[2], y[3]);
            if (out_f32) reinterpret_cast<float4*>(out_f32)[idx] = o;
            if (q0) reinterpret_cast<uint32_t*>(q0)[idx] = quant4(o, qa);
            if (q1) reinterpret_cast<uint32_t*>(q1)[idx] = quant4(o, qb);
            if (q2) reinterpret_cast<uint32_t*>(q2)[idx] = quant4(o, qc);
            if (qraw) reinterpret_cast<uint32_t*>(qraw)[idx] = quant4(v, qr);
        }
        if (Q <= 256) break;
    }
}
// The common sampling form -- ONE int8 operand out (+ optionally the raw-input operand), no fp32 output, no scale-shift,
// channel counts in multiples of 16 -- with 16 channels per thread: four 16-byte loads in flight per lane and one 16-byte
// store per operand (k_gn_apply moves 16 B in / 4 B out per lane and iteration).  Same arithmetic per 4-channel group as
// k_gn_apply (same `quant4`, same boundary test granularity): identical bits.
__global__ void __launch_bounds__(256) k_gn_apply16(const float* __restrict__ x, const float* __restrict__ x2, int64_t C1,
                                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int64_t HW, int64_t C, int64_t G, int silu,
                                                    int8_t* __restrict__ q0, const QP* __restrict__ qp, int rows_per_block,
                                                    int8_t* __restrict__ qraw, const QP* __restrict__ qpr, int64_t raw_split,
                                                    int64_t B2) {
    const int64_t b = blockIdx.y;
    const int64_t b2 = B2 > 0 ? b % B2 : b;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < HW ? r0 + rows_per_block : HW;
    const int G16 = (int)(C >> 4), RS = 256 / G16;
    const int tid = threadIdx.x;
    const int grp = tid % G16, rs = tid / G16;
    if (rs >= RS) return;
    const int cpg = (int)(C / G);
    const QP qa = qp_load(qp, 0);
    float a[16], bb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = grp * 16 + j;
        const int g = c / cpg;
        const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
        a[j] = rstd * gamma[c];
        bb[j] = beta[c] - mean * a[j];
    }
    const bool first = grp * 16 < C1;
    const int64_t Ca = first ? C1 : C - C1;
    const float* src = first ? x + grp * 16 : x2 + (grp * 16 - C1);
    const int64_t bs = first ? b : b2;
    QP qr = qa;
    if (qraw) qr = qp_load(qpr, (raw_split > 0 && grp * 16 >= raw_split) ? 1 : 0);
    for (int64_t r = r0 + rs; r < r1; r += RS) {
        const float4* p = reinterpret_cast<const float4*>(src + (bs * HW + r) * Ca);
        const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        const float4 vv[4] = {v0, v1, v2, v3};
        uint4 w, wr;
        uint32_t* wp = reinterpret_cast<uint32_t*>(&w);
        uint32_t* wrp = reinterpret_cast<uint32_t*>(&wr);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = vv[k];
            float y[4] = {v.x * a[4 * k] + bb[4 * k], v.y * a[4 * k + 1] + bb[4 * k + 1], v.z * a[4 * k + 2] + bb[4 * k + 2],
                          v.w * a[4 * k + 3] + bb[4 * k + 3]};
            if (silu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = silu_rcp(y[j]);
            }
            wp[k] = quant4(make_float4(y[0], y[1], y[2], y[3]), qa);
            if (qraw) wrp[k] = quant4(v, qr);
        }
        const int64_t o = (b * HW + r) * C + grp * 16;
        *reinterpret_cast<uint4*>(q0 + o) = w;
        if (qraw) *reinterpret_cast<uint4*>(qraw + o) = wr;
    }
}

extern "C" int edadm_groupnorm_apply_cat_raw(const float* x1, int64_t C1, const float* x2, int64_t C2, const float* stats,
                                             const float* gamma, const float* beta, const float* scale_shift, int64_t B,
                                             int64_t HW, int64_t G, int silu, float* out_f32, int8_t* q0, int8_t* q1,
                                             int8_t* q2, const float* qp, int nq, int8_t* qraw, const float* qp_raw,
                                             int64_t raw_split, int64_t B2, void* stream) {
    const int64_t C = C1 + (x2 ? C2 : 0);
    if (!x1 || !stats || !gamma || !beta || B <= 0 || HW <= 0 || C1 <= 0 || (x2 && C2 <= 0) || (C & 3) || (C1 & 3) || (C % G))
        return EDADM_EINVAL;
    if ((q0 || q1 || q2) && !qp) return EDADM_EINVAL;
    if (qraw && (!qp_raw || raw_split < 0 || raw_split >= C || (raw_split & 3))) return EDADM_EINVAL;
    if (B2 < 0 || (B2 > 0 && (!x2 || B % B2))) return EDADM_EINVAL;
    if (!out_f32 && !scale_shift && nq == 1 && q0 && !q1 && !q2 && (C & 15) == 0 && (C1 & 15) == 0 && C <= 4096 &&
        (!qraw || (raw_split & 15) == 0) && !((uintptr_t)q0 & 15) && !((uintptr_t)qraw & 15) && !((uintptr_t)x1 & 15) &&
        !((uintptr_t)x2 & 15)) {
        const int G16 = (int)(C >> 4), RS = 256 / G16;
        int rpb16 = (int)(16384 / C);
        rpb16 = rpb16 < RS ? RS : (rpb16 / RS) * RS;
        const unsigned gx16 = (unsigned)((HW + rpb16 - 1) / rpb16);
        hipLaunchKernelGGL(k_gn_apply16, dim3(gx16, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x1, x2, C1, stats, gamma, beta,
                           HW, C, G, silu, q0, (const QP*)qp, rpb16, qraw, (const QP*)qp_raw, raw_split, B2);
        return edadm_launch_status();
    }
    // rows per block: a multiple of what keeps (quad) fixed per thread when Q | 256, ~16 KB of input per block
    int rpb = (int)(16384 / C);   // ~64 KB of fp32 input per block
    if (rpb < 1) rpb = 1;
    const unsigned gx = (unsigned)((HW + rpb - 1) / rpb);
    hipLaunchKernelGGL(k_gn_apply, dim3(gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x1, x2, C1, stats, gamma, beta,
                       scale_shift, HW, C, G, silu, out_f32, q0, q1, q2, (const QP*)qp, nq, rpb, qraw, (const QP*)qp_raw,
                       raw_split, B2);
    return edadm_launch_status();
}
extern "C" int edadm_groupnorm_apply_cat(const float* x1, int64_t C1, const float* x2, int64_t C2, const float* stats,
                                         const float* gamma, const float* beta, const float* scale_shift, int64_t B,
                                         int64_t HW, int64_t G, int silu, float* out_f32, int8_t* q0, int8_t* q1,
                                         int8_t* q2, const float* qp, int nq, void* stream) {
    return edadm_groupnorm_apply_cat_raw(x1, C1, x2, C2, stats, gamma, beta, scale_shift, B, HW, G, silu, out_f32, q0, q1, q2,
                                         qp, nq, nullptr, nullptr, 0, 0, stream);
}
extern "C" int edadm_groupnorm_apply(const float* x, const float* stats, const float* gamma, const float* beta,
                                     const float* scale_shift, int64_t B, int64_t HW, int64_t C, int64_t G,
                                     int silu, float* out_f32, int8_t* q0, int8_t* q1, int8_t* q2, const float* qp,
                                     int nq, void* stream) {
    return edadm_groupnorm_apply_cat(x, C, nullptr, 0, stats, gamma, beta, scale_shift, B, HW, G, silu, out_f32, q0, q1, q2,
                                     qp, nq, stream);
}

// ------------------------------------------------------------------ LayerNorm (+ up to 3 quantised outputs)
// one wave per row; the row stays in registers (C <= 64*MAXPL) between the two passes
#define LN_MAXPL 32
#define LN_MAXV4 8
// C % 4 == 0 (every layer of the UNets): a lane owns float4 chunks lane, lane+64, ... -> 16-byte loads, one packed
// dword per quantised output
// sum over the LPR lanes that share a row (LPR = 64: the wave; 32: each half on its own)
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// LPR lanes per row: 64 = one row per wave; 32 = two rows per wave, one per half -- a 384-wide row is 96 chunks: 3 x 32 lanes
// exactly, against 2 x 64 with a quarter of the lanes idle (and twice the bytes in flight per wave)
template <int LPR>
__global__ void __launch_bounds__(256) k_ln_quant_v4(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int64_t rows, int64_t C,
                                                     float eps, float* __restrict__ out_f32, int8_t* __restrict__ q0,
                                                     int8_t* __restrict__ q1, int8_t* __restrict__ q2,
                                                     const QP* __restrict__ qp, int nq,
                                                     const float* __restrict__ radd, int64_t rows_per_batch,
                                                     int64_t xrows, float* __restrict__ sum_out) {
    // radd: the input of the norm is x[row] + radd[row / rows_per_batch] (edadm_add_rowbcast folded in: the sum is
    // written to sum_out, the updated residual stream); xrows > 0: x has xrows < rows rows, read periodically
    constexpr int RPW = 64 / LPR;                        // rows per wave
    const int lane = threadIdx.x & (LPR - 1);
    const int64_t row_ = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LPR);
    const bool live = row_ < rows;                       // a dead half still takes part in the shuffles of its wave
    const int64_t row = live ? row_ : rows - 1;
    QP qa = qp ? qp_load(qp, 0) : QP{1, 0, 255, 1}, qb = (qp && nq > 1) ? qp_load(qp, 1) : qa, qc = (qp && nq > 2) ? qp_load(qp, 2) : qa;
    const int Q = (int)(C >> 2);
    const float4* xr = reinterpret_cast<const float4*>(x + (xrows > 0 ? row % xrows : row) * C);
    float4 v[LN_MAXV4];
    float s = 0.f;
    if (radd) {
        const float4* rr = reinterpret_cast<const float4*>(radd + (row / rows_per_batch) * C);
#pragma unroll
        for (int j = 0; j < LN_MAXV4; ++j) {
            const int c = j * LPR + lane;
            if (c < Q) {
                const float4 a = xr[c], b = rr[c];
                v[j] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
                if (live) reinterpret_cast<float4*>(sum_out)[row * Q + c] = v[j];
            } else {
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < LN_MAXV4; ++j) {
            const int c = j * LPR + lane;
            v[j] = c < Q ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < LN_MAXV4; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = row_sum<LPR>(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV4; ++j) {
        const int c = j * LPR + lane;
        if (c < Q) {
            const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
            ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float rstd = 1.0f / sqrtf(row_sum<LPR>(ss) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < LN_MAXV4; ++j) {
        const int c = j * LPR + lane;
        if (c < Q && live) {
            const float4 g4 = reinterpret_cast<const float4*>(gamma)[c], b4 = reinterpret_cast<const float4*>(beta)[c];
            float4 y;
            y.x = (v[j].x - mean) * rstd * g4.x + b4.x;
            y.y = (v[j].y - mean) * rstd * g4.y + b4.y;
            y.z = (v[j].z - mean) * rstd * g4.z + b4.z;
            y.w = (v[j].w - mean) * rstd * g4.w + b4.w;
            const int64_t o = row * Q + c;
            if (out_f32) reinterpret_cast<float4*>(out_f32)[o] = y;
            if (q0) reinterpret_cast<uint32_t*>(q0)[o] = quant4(y, qa);
            if (q1) reinterpret_cast<uint32_t*>(q1)[o] = quant4(y, qb);
            if (q2) reinterpret_cast<uint32_t*>(q2)[o] = quant4(y, qc);
        }
    }
}
__global__ void __launch_bounds__(256) k_ln_quant(const float* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int64_t rows, int64_t C,
                                                  float eps, float* __restrict__ out_f32, int8_t* __restrict__ q0,
                                                  int8_t* __restrict__ q1, int8_t* __restrict__ q2,
                                                  const QP* __restrict__ qp, int nq) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    QP qa = qp ? qp_load(qp, 0) : QP{1, 0, 255, 1}, qb = (qp && nq > 1) ? qp_load(qp, 1) : qa, qc = (qp && nq > 2) ? qp_load(qp, 2) : qa;
    const float* xr = x + row * C;
    float v[LN_MAXPL];
    const int npl = (int)((C + 63) / 64);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXPL; ++j) {
        if (j < npl) {
            const int64_t c = (int64_t)j * 64 + lane;
            v[j] = c < C ? xr[c] : 0.f;
            s += v[j];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXPL; ++j) {
        if (j < npl) {
            const int64_t c = (int64_t)j * 64 + lane;
            const float d = c < C ? v[j] - mean : 0.f;
            ss += d * d;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < LN_MAXPL; ++j) {
        if (j < npl) {
            const int64_t c = (int64_t)j * 64 + lane;
            if (c < C) {
                const float y = (v[j] - mean) * rstd * gamma[c] + beta[c];
                if (out_f32) out_f32[row * C + c] = y;
                if (q0) q0[row * C + c] = (int8_t)q_code_i8(y, qa);
                if (q1) q1[row * C + c] = (int8_t)q_code_i8(y, qb);
                if (q2) q2[row * C + c] = (int8_t)q_code_i8(y, qc);
            }
        }
    }
}
// two rows per wave when 32-lane rows leave fewer idle lane slots than 64-lane rows (C = 384: 0 against 32 of 128)
static bool ln_half_rows(int64_t C) {
    const int64_t Q = C >> 2;
    return Q <= 32 * LN_MAXV4 && (Q + 31) / 32 * 32 - Q < (Q + 63) / 64 * 64 - Q;
}
extern "C" int edadm_layernorm_quant(const float* x, const float* gamma, const float* beta, int64_t rows, int64_t C,
                                     float eps, float* out_f32, int8_t* q0, int8_t* q1, int8_t* q2, const float* qp,
                                     int nq, void* stream) {
    if (!x || !gamma || !beta || rows <= 0 || C <= 0 || C > 64 * LN_MAXPL) return EDADM_EINVAL;
    if ((q0 || q1 || q2) && !qp) return EDADM_EINVAL;
    const bool al = !(((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out_f32) & 15) &&
                    !(((uintptr_t)q0 | (uintptr_t)q1 | (uintptr_t)q2) & 3);
    if ((C & 3) == 0 && C <= 256 * LN_MAXV4 && al) {
        if (ln_half_rows(C))
            hipLaunchKernelGGL(k_ln_quant_v4<32>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                               beta, rows, C, eps, out_f32, q0, q1, q2, (const QP*)qp, nq, (const float*)nullptr, (int64_t)1,
                               (int64_t)0, (float*)nullptr);
        else
            hipLaunchKernelGGL(k_ln_quant_v4<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                               beta, rows, C, eps, out_f32, q0, q1, q2, (const QP*)qp, nq, (const float*)nullptr, (int64_t)1,
                               (int64_t)0, (float*)nullptr);
    } else
        hipLaunchKernelGGL(k_ln_quant, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                           beta, rows, C, eps, out_f32, q0, q1, q2, (const QP*)qp, nq);
    return edadm_launch_status();
}

// LayerNorm of x[row] + radd[row / rows_per_batch] with the sum written out as the updated residual stream: the
// broadcast add of a per-image vector (edadm_add_rowbcast) folded into the norm that reads its result next
extern "C" int edadm_layernorm_quant_radd(const float* x, int64_t xrows, const float* radd, int64_t rows_per_batch,
                                          float* sum_out, const float* gamma, const float* beta, int64_t rows, int64_t C,
                                          float eps, int8_t* q0, int8_t* q1, int8_t* q2, const float* qp, int nq,
                                          void* stream) {
    if (!x || !radd || !sum_out || !gamma || !beta || rows <= 0 || C <= 0 || (C & 3) || C > 256 * LN_MAXV4 ||
        rows_per_batch <= 0 || xrows < 0 || (xrows > 0 && rows % xrows))
        return EDADM_EINVAL;
    if ((q0 || q1 || q2) && !qp) return EDADM_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)radd | (uintptr_t)sum_out | (uintptr_t)gamma | (uintptr_t)beta) & 15) ||
        (((uintptr_t)q0 | (uintptr_t)q1 | (uintptr_t)q2) & 3))
        return EDADM_EINVAL;
    if (ln_half_rows(C))
        hipLaunchKernelGGL(k_ln_quant_v4<32>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                           rows, C, eps, (float*)nullptr, q0, q1, q2, (const QP*)qp, nq, radd, rows_per_batch,
                           xrows == rows ? (int64_t)0 : xrows, sum_out);
    else
        hipLaunchKernelGGL(k_ln_quant_v4<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                           rows, C, eps, (float*)nullptr, q0, q1, q2, (const QP*)qp, nq, radd, rows_per_batch,
                           xrows == rows ? (int64_t)0 : xrows, sum_out);
    return edadm_launch_status();
}

// ------------------------------------------------------------------ small fused producers
__global__ void __launch_bounds__(256) k_silu_q(const float* __restrict__ x, int8_t* __restrict__ out, int64_t n,
                                                const QP* __restrict__ qp) {
    const QP q = qp_load(qp, 0);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (int8_t)q_code_i8(silu_rcp(x[i]), q);
}
extern "C" int edadm_silu_quant_i8(const float* x, int8_t* out, int64_t n, const float* qp, void* stream) {
    if (!x || !out || !qp || n <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_silu_q, dim3(edadm_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n,
                       (const QP*)qp);
    return edadm_launch_status();
}
__global__ void __launch_bounds__(256) k_silu(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = silu_f(x[i]);
}
extern "C" int edadm_silu(const float* x, float* out, int64_t n, void* stream) {
    if (!x || !out || n <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_silu, dim3(edadm_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    return edadm_launch_status();
}
// GEGLU: x[rows][2*inner] -> a * gelu(gate) quantised  (attention.py:37-45; exact-erf gelu)
__global__ void __launch_bounds__(256) k_geglu_q(const float* __restrict__ x, int8_t* __restrict__ out,
                                                 int64_t rows, int64_t inner, const QP* __restrict__ qp) {
    const QP q = qp_load(qp, 0);
    const int64_t n = rows * inner, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / inner, c = i - r * inner;
        const float a = x[r * 2 * inner + c], g = x[r * 2 * inner + inner + c];
        const float gl = 0.5f * g * (1.0f + erf_fast(g * 0.70710678118654752440f));
        out[i] = (int8_t)q_code_i8(a * gl, q);
    }
}
extern "C" int edadm_geglu_quant_i8(const float* x, int8_t* out, int64_t rows, int64_t inner, const float* qp,
                                    void* stream) {
    if (!x || !out || !qp || rows <= 0 || inner <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_geglu_q, dim3(edadm_grid(rows * inner, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                       rows, inner, (const QP*)qp);
    return edadm_launch_status();
}
__global__ void __launch_bounds__(256) k_add(const float* __restrict__ a, const float* __restrict__ b,
                                             float* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}
extern "C" int edadm_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_add, dim3(edadm_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return edadm_launch_status();
}
// out[m][c] = x[m][c] + r[m / rows_per_batch][c]: a per-image vector added to every token of the image (cross-attention
// over a one-token context is the same vector for every query: softmax over one key is 1)
__global__ void __launch_bounds__(256) k_add_rowbcast(const float* __restrict__ x, const float* __restrict__ r,
                                                      float* __restrict__ out, int64_t rows, int64_t C4,
                                                      int64_t rows_per_batch, int64_t xrows) {
    // xrows > 0: x holds xrows < rows rows and is read periodically (the half of a guidance pair that both halves share)
    const int64_t n = rows * C4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t m = i / C4, c = i - m * C4;
        const float4 a = reinterpret_cast<const float4*>(x)[xrows > 0 ? (m % xrows) * C4 + c : i];
        const float4 b = reinterpret_cast<const float4*>(r)[(m / rows_per_batch) * C4 + c];
        reinterpret_cast<float4*>(out)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
extern "C" int edadm_add_rowbcast_rep(const float* x, const float* r, float* out, int64_t rows, int64_t C,
                                      int64_t rows_per_batch, int64_t xrows, void* stream) {
    if (!x || !r || !out || rows <= 0 || C <= 0 || (C & 3) || rows_per_batch <= 0 || xrows < 0 || (xrows > 0 && rows % xrows))
        return EDADM_EINVAL;
    hipLaunchKernelGGL(k_add_rowbcast, dim3(edadm_grid(rows * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, r, out,
                       rows, C / 4, rows_per_batch, xrows);
    return edadm_launch_status();
}
extern "C" int edadm_add_rowbcast(const float* x, const float* r, float* out, int64_t rows, int64_t C,
                                  int64_t rows_per_batch, void* stream) {
    return edadm_add_rowbcast_rep(x, r, out, rows, C, rows_per_batch, 0, stream);
}
// channel concat of two NHWC tensors (the UNet skip connection, openaimodel.py:778)
__global__ void __launch_bounds__(256) k_concat(const float* __restrict__ a, int64_t Ca, const float* __restrict__ b,
                                                int64_t Cb, float* __restrict__ out, int64_t rows) {
    const int64_t C4 = (Ca + Cb) >> 2, A4 = Ca >> 2, B4 = Cb >> 2;
    const int64_t n = rows * C4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / C4, c = i - r * C4;
        reinterpret_cast<float4*>(out)[i] = c < A4 ? reinterpret_cast<const float4*>(a)[r * A4 + c]
                                                   : reinterpret_cast<const float4*>(b)[r * B4 + (c - A4)];
    }
}
extern "C" int edadm_concat_c(const float* a, int64_t Ca, const float* b, int64_t Cb, float* out, int64_t rows,
                              void* stream) {
    if (!a || !b || !out || rows <= 0 || (Ca & 3) || (Cb & 3)) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_concat, dim3(edadm_grid(rows * (Ca + Cb) / 4, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       Ca, b, Cb, out, rows);
    return edadm_launch_status();
}
__global__ void __launch_bounds__(256) k_avgpool2(const float* __restrict__ x, float* __restrict__ out, int64_t B,
                                                  int64_t H, int64_t W, int64_t C) {
    const int64_t Ho = H / 2, Wo = W / 2, n = B * Ho * Wo * C, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t c = i % C, p = i / C, xo = p % Wo, yo = (p / Wo) % Ho, b = p / (Wo * Ho);
        const float* s = x + ((b * H + 2 * yo) * W + 2 * xo) * C + c;
        out[i] = (s[0] + s[C] + s[W * C] + s[W * C + C]) * 0.25f;
    }
}
extern "C" int edadm_avgpool2_nhwc(const float* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C,
                                   void* stream) {
    if (!x || !out || B <= 0 || H < 2 || W < 2 || C <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_avgpool2, dim3(edadm_grid(B * H * W * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       out, B, H, W, C);
    return edadm_launch_status();
}
__global__ void __launch_bounds__(256) k_upsample2(const float* __restrict__ x, float* __restrict__ out, int64_t B,
                                                   int64_t H, int64_t W, int64_t C) {
    const int64_t Ho = H * 2, Wo = W * 2, n = B * Ho * Wo * C, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t c = i % C, p = i / C, xo = p % Wo, yo = (p / Wo) % Ho, b = p / (Wo * Ho);
        out[i] = x[((b * H + yo / 2) * W + xo / 2) * C + c];
    }
}
extern "C" int edadm_upsample2_nhwc(const float* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C,
                                    void* stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_upsample2, dim3(edadm_grid(B * H * W * C * 4, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       out, B, H, W, C);
    return edadm_launch_status();
}

// ------------------------------------------------------------------ K6: softmax rows + quantise to f16 codes
// cols % 4 == 0, cols <= 4096: the row lives in registers (one read of the scores), exp once per element,
// the probability's code from one multiply (exact division only next to a rounding boundary), 8-byte stores
#define SM_MAXV4 16
__global__ void __launch_bounds__(256) k_softmax_q_v4(const float* __restrict__ s, __half* __restrict__ out,
                                                      int64_t rows, int64_t cols, int64_t ldo,
                                                      const QP* __restrict__ qp) {
    const QP q = qp_load(qp, 0);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Q = (int)(cols >> 2);
    const float4* sr = reinterpret_cast<const float4*>(s + row * cols);
    float4 v[SM_MAXV4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SM_MAXV4; ++j) {
        const int c = j * 64 + lane;
        if (c < Q) {
            v[j] = sr[c];
            mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < SM_MAXV4; ++j) {
        const int c = j * 64 + lane;
        if (c < Q) {
            v[j].x = exp_hw(v[j].x - mx); v[j].y = exp_hw(v[j].y - mx); v[j].z = exp_hw(v[j].z - mx); v[j].w = exp_hw(v[j].w - mx);
            sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / (sum * q.d);                  // e / sum / delta ~ e * inv; boundary cases redo both divisions
    __half* orow = out + row * ldo;
#pragma unroll
    for (int j = 0; j < SM_MAXV4; ++j) {
        const int c = j * 64 + lane;
        if (c < Q) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            float r[4];
            bool near = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = e[k] * inv;
                r[k] = rintf(t);
                near |= fmaf(t, 3.6e-7f, fabsf(t - r[k])) > 0.5f - 4e-5f;      // relative band (attn.hip, common.h)
            }
            if (__builtin_expect(near, 0)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("" : "+v"(r[k]));
                    r[k] = rintf((e[k] / sum) / q.d);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fminf(fmaxf(r[k] + q.z, 0.f), q.qmax) - q.z;
            __half2 h0 = __floats2half2_rn(r[0], r[1]), h1 = __floats2half2_rn(r[2], r[3]);
            uint2 pk;
            pk.x = *reinterpret_cast<uint32_t*>(&h0);
            pk.y = *reinterpret_cast<uint32_t*>(&h1);
            *reinterpret_cast<uint2*>(orow + 4 * c) = pk;
        }
    }
    for (int64_t c = cols + lane; c < ldo; c += 64) orow[c] = __float2half(0.f);
}
__global__ void __launch_bounds__(256) k_softmax_q(const float* __restrict__ s, __half* __restrict__ out,
                                                   int64_t rows, int64_t cols, int64_t ldo,
                                                   const QP* __restrict__ qp) {
    const QP q = qp_load(qp, 0);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * cols;
    float mx = -INFINITY;
    for (int64_t c = lane; c < cols; c += 64) mx = fmaxf(mx, sr[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int64_t c = lane; c < cols; c += 64) sum += exp_hw(sr[c] - mx);
    sum = wave_sum(sum);
    for (int64_t c = lane; c < cols; c += 64) {
        const float p = exp_hw(sr[c] - mx) / sum;
        const float code = q_code_f(p, q);
        out[row * ldo + c] = __float2half(code - q.z);
    }
    for (int64_t c = cols + lane; c < ldo; c += 64) out[row * ldo + c] = __float2half(0.f);
}
extern "C" int edadm_softmax_quant_f16(const float* s, void* out, int64_t rows, int64_t cols, int64_t ldo,
                                       const float* qp, void* stream) {
    if (!s || !out || !qp || rows <= 0 || cols <= 0 || ldo < cols) return EDADM_EINVAL;
    if ((cols & 3) == 0 && (ldo & 3) == 0 && cols <= 256 * SM_MAXV4 && !((uintptr_t)s & 15) && !((uintptr_t)out & 7))
        hipLaunchKernelGGL(k_softmax_q_v4, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s,
                           (__half*)out, rows, cols, ldo, (const QP*)qp);
    else
        hipLaunchKernelGGL(k_softmax_q, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s,
                           (__half*)out, rows, cols, ldo, (const QP*)qp);
    return edadm_launch_status();
}

// plain fp32 row softmax (the first-stage decoder's attention block, model.py:193-195): register-resident row
__global__ void __launch_bounds__(256) k_softmax_f32(const float* __restrict__ s, float* __restrict__ out, int64_t rows,
                                                     int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Q = (int)(cols >> 2);
    const float4* sr = reinterpret_cast<const float4*>(s + row * cols);
    float4 v[SM_MAXV4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SM_MAXV4; ++j) {
        const int c = j * 64 + lane;
        if (c < Q) {
            v[j] = sr[c];
            mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < SM_MAXV4; ++j) {
        const int c = j * 64 + lane;
        if (c < Q) {
            v[j].x = expf(v[j].x - mx); v[j].y = expf(v[j].y - mx); v[j].z = expf(v[j].z - mx); v[j].w = expf(v[j].w - mx);
            sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int j = 0; j < SM_MAXV4; ++j) {
        const int c = j * 64 + lane;
        if (c < Q) {
            float4 o;
            o.x = v[j].x / sum; o.y = v[j].y / sum; o.z = v[j].z / sum; o.w = v[j].w / sum;
            reinterpret_cast<float4*>(out + row * cols)[c] = o;
        }
    }
}
extern "C" int edadm_softmax_f32(const float* s, float* out, int64_t rows, int64_t cols, void* stream) {
    if (!s || !out || rows <= 0 || cols <= 0 || (cols & 3) || cols > 256 * SM_MAXV4 || ((uintptr_t)s & 15) || ((uintptr_t)out & 15))
        return EDADM_EINVAL;
    hipLaunchKernelGGL(k_softmax_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, out, rows, cols);
    return edadm_launch_status();
}

// f16 [b][n][d] -> [b][d][n]  (B operand of the PV product); pads n up to ldo with zeros
__global__ void __launch_bounds__(256) k_transpose_f16(const __half* __restrict__ x, int64_t ldx, int64_t strideX,
                                                       __half* __restrict__ out, int64_t ldo, int64_t strideO,
                                                       int64_t n, int64_t d) {
    __shared__ __half tile[32][34];
    const int64_t b = blockIdx.z;
    const int64_t n0 = (int64_t)blockIdx.x * 32, d0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int64_t nn = n0 + j, dd = d0 + tx;
        tile[j][tx] = (nn < n && dd < d) ? x[b * strideX + nn * ldx + dd] : __float2half(0.f);
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t dd = d0 + j, nn = n0 + tx;
        if (dd < d && nn < ldo) out[b * strideO + dd * ldo + nn] = tile[tx][j];
    }
}
extern "C" int edadm_transpose_f16(const void* x, int64_t ldx, int64_t strideX, void* out, int64_t ldo,
                                   int64_t strideO, int64_t batch, int64_t n, int64_t d, void* stream) {
    if (!x || !out || batch <= 0 || n <= 0 || d <= 0 || ldo < n) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_transpose_f16, dim3((unsigned)((ldo + 31) / 32), (unsigned)((d + 31) / 32), (unsigned)batch),
                       dim3(256), 0, (hipStream_t)stream, (const __half*)x, ldx, strideX, (__half*)out, ldo, strideO,
                       n, d);
    return edadm_launch_status();
}

// int4 nibble codes (two per byte, low nibble first) -> int8 operand code - zp[row]
__global__ void __launch_bounds__(256) k_unpack_w4(const uint8_t* __restrict__ p, const float* __restrict__ zp,
                                                   int8_t* __restrict__ out, int64_t rows, int64_t cols) {
    const int64_t n = rows * cols, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / cols;
        const uint8_t byte = p[i >> 1];
        const int code = (i & 1) ? (byte >> 4) : (byte & 15);
        out[i] = (int8_t)(code - (int)zp[r]);
    }
}
extern "C" int edadm_unpack_w4(const uint8_t* packed, const float* zp, int8_t* out, int64_t rows, int64_t cols,
                               void* stream) {
    if (!packed || !zp || !out || rows <= 0 || cols <= 0 || ((rows * cols) & 1)) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_unpack_w4, dim3(edadm_grid(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, packed,
                       zp, out, rows, cols);
    return edadm_launch_status();
}

// int8 operand (code - zp_row) -> two 4-bit codes per byte (low nibble = even element): the storage format of
// calibrated W4 weights (edadm/state.py); the inverse of k_unpack_w4
__global__ void __launch_bounds__(256) k_pack_w4(const int8_t* __restrict__ w, const float* __restrict__ zp,
                                                 uint8_t* __restrict__ out, int64_t rows, int64_t cols) {
    const int64_t nb = (rows * cols) >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) {
        const int64_t e0 = 2 * i, e1 = 2 * i + 1;
        const int c0 = (int)w[e0] + (int)zp[e0 / cols], c1 = (int)w[e1] + (int)zp[e1 / cols];
        out[i] = (uint8_t)((c0 & 15) | ((c1 & 15) << 4));
    }
}
extern "C" int edadm_pack_w4(const int8_t* w, const float* zp, uint8_t* packed, int64_t rows, int64_t cols,
                             void* stream) {
    if (!w || !zp || !packed || rows <= 0 || cols <= 0 || ((rows * cols) & 1)) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_pack_w4, dim3(edadm_grid(rows * cols / 2, 256)), dim3(256), 0, (hipStream_t)stream, w, zp, packed,
                       rows, cols);
    return edadm_launch_status();
}

// ------------------------------------------------------------------ fp32 im2col / col2im / slab sum (H1 contraction)
// NHWC x[B][H][W][C] -> cols[m][(ky*KW+kx)*C + c], m = (b, y, x) over the Ho x Wo outputs; zero padding.
__global__ void __launch_bounds__(256) k_im2col_f32(const float* __restrict__ x, float* __restrict__ cols, int64_t B,
                                                    int64_t H, int64_t W, int64_t C, int64_t Ho, int64_t Wo, int KH,
                                                    int KW, int stride, int pad) {
    const int64_t C4 = C >> 2, K4 = (int64_t)KH * KW * C4;
    const int64_t n = B * Ho * Wo * K4, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        const int64_t m = i / K4, k = i - m * K4;
        const int tap = (int)(k / C4);
        const int64_t c4 = k - (int64_t)tap * C4;
        const int64_t b = m / (Ho * Wo), r = m - b * Ho * Wo, y = r / Wo, xx = r - y * Wo;
        const int64_t iy = y * stride + tap / KW - pad, ix = xx * stride + tap % KW - pad;
        float4 v = make_float4(0, 0, 0, 0);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = reinterpret_cast<const float4*>(x)[((b * H + iy) * W + ix) * C4 + c4];
        reinterpret_cast<float4*>(cols)[i] = v;
    }
}
extern "C" int edadm_im2col_f32(const float* x, float* cols, int64_t B, int64_t H, int64_t W, int64_t C, int64_t Ho,
                                int64_t Wo, int KH, int KW, int stride, int pad, void* stream) {
    if (!x || !cols || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || Ho <= 0 || Wo <= 0 || KH < 1 || KW < 1 ||
        stride < 1 || pad < 0)
        return EDADM_EINVAL;
    hipLaunchKernelGGL(k_im2col_f32, dim3(edadm_grid(B * Ho * Wo * KH * KW * C / 4, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, cols, B, H, W, C, Ho, Wo, KH, KW, stride, pad);
    return edadm_launch_status();
}
// gather form of the adjoint (deterministic): dx[b][iy][ix][c] = sum over taps of dcols at the output pixel that read it
__global__ void __launch_bounds__(256) k_col2im_f32(const float* __restrict__ dcols, float* __restrict__ dx, int64_t B,
                                                    int64_t H, int64_t W, int64_t C, int64_t Ho, int64_t Wo, int KH,
                                                    int KW, int stride, int pad) {
    const int64_t C4 = C >> 2, K4 = (int64_t)KH * KW * C4;
    const int64_t n = B * H * W * C4, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        const int64_t c4 = i % C4, p = i / C4, ix = p % W, iy = (p / W) % H, b = p / (W * H);
        float4 acc = make_float4(0, 0, 0, 0);
        for (int ky = 0; ky < KH; ++ky) {
            const int64_t ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int64_t y = ty / stride;
            if (y >= Ho) continue;
            for (int kx = 0; kx < KW; ++kx) {
                const int64_t tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int64_t xo = tx / stride;
                if (xo >= Wo) continue;
                const float4 v = reinterpret_cast<const float4*>(dcols)[((b * Ho + y) * Wo + xo) * K4 + (ky * KW + kx) * C4 + c4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        reinterpret_cast<float4*>(dx)[i] = acc;
    }
}
extern "C" int edadm_col2im_f32(const float* dcols, float* dx, int64_t B, int64_t H, int64_t W, int64_t C, int64_t Ho,
                                int64_t Wo, int KH, int KW, int stride, int pad, void* stream) {
    if (!dcols || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || Ho <= 0 || Wo <= 0 || KH < 1 || KW < 1 ||
        stride < 1 || pad < 0)
        return EDADM_EINVAL;
    hipLaunchKernelGGL(k_col2im_f32, dim3(edadm_grid(B * H * W * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, dcols,
                       dx, B, H, W, C, Ho, Wo, KH, KW, stride, pad);
    return edadm_launch_status();
}
// out[i] = sum_s slabs[s][i] in slab order (split-K partials of the weight gradient; deterministic)
__global__ void __launch_bounds__(256) k_sum_slabs(const float* __restrict__ slabs, float* __restrict__ out, int64_t n,
                                                   int64_t S) {
    const int64_t st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
        float a = 0.f;
        for (int64_t s = 0; s < S; ++s) a += slabs[s * n + i];
        out[i] = a;
    }
}
extern "C" int edadm_sum_slabs(const float* slabs, float* out, int64_t n, int64_t S, void* stream) {
    if (!slabs || !out || n <= 0 || S <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_sum_slabs, dim3(edadm_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, slabs, out, n, S);
    return edadm_launch_status();
}
