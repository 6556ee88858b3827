// H1 (reconstruction loop) normalisation / activation / softmax kernels: forward AND backward of the non-contraction ops of
// the calibration graph, on the reference's tensor layouts (NCHW feature maps, [rows][C] token matrices), fp32.
//
//   GroupNorm (+ SiLU)   ddim/models/diffusion.py:27-35, openaimodel.py:215-223, quant_block.py:86-116,321-348
//   LayerNorm            ldm/modules/attention.py:201-203,277-285
//   GEGLU                ldm/modules/attention.py:37-45
//   SiLU                 time-embedding paths (diffusion.py:322-324, openaimodel.py:233-240)
//   softmax              quant_block.py:204-235,427-446
//
// These are what autograd runs between the fake-quant (K1), AdaRound (K2) and contraction (K11) kernels of a
// reconstruction iteration.  Only input gradients are produced: the loop trains AdaRound alphas and activation step
// sizes, never the normalisation affines (block_recon.py:44-108), so their gradients are not computed at all.
// All kernels are HBM-bound: algorithmic bytes = 4 B x (elements read + written), the statistics pass re-reads x.
#include "common.h"
#include <type_traits>
#include "../../include/edadm.h"

// ------------------------------------------------------------------------------------------------ block reductions
__device__ __forceinline__ void block_sum2_d(double& a, double& b, double* sm /* [2 * waves] */) {
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) { sm[2 * w] = a; sm[2 * w + 1] = b; }
    __syncthreads();
    double ra = 0, rb = 0;
    for (int i = 0; i < nw; ++i) { ra += sm[2 * i]; rb += sm[2 * i + 1]; }
    a = ra;
    b = rb;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ GroupNorm (+ SiLU), NCHW
// One workgroup per (image, group): the group's channels are one contiguous run of cg * HW floats.  Statistics in fp64
// (one pass, sum and sum of squares: exact enough that mean / rstd agree with a two-pass fp32 evaluation to the last
// bits), then the apply pass re-reads the run (L2 / Infinity Cache resident: <= 1.5 MB per group at 64x64).
__global__ void __launch_bounds__(512) k_gn_fwd_nchw(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ stats, int64_t C, int64_t HW, int G, float eps,
                                                     int silu) {
    __shared__ double sm[16];
    const int64_t bg = blockIdx.x;
    const int g = (int)(bg % G);
    const int64_t cg = C / G, n = cg * HW;
    const float* xr = x + bg * n;
    float* yr = y + bg * n;
    double s = 0, ss = 0;
    const bool v4 = (HW & 3) == 0;
    if (v4) {
        const float4* x4 = reinterpret_cast<const float4*>(xr);
        for (int64_t i = threadIdx.x; i < (n >> 2); i += blockDim.x) {
            const float4 v = x4[i];
            s += ((double)v.x + v.y) + ((double)v.z + v.w);
            ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { const double v = xr[i]; s += v; ss += v * v; }
    }
    block_sum2_d(s, ss, sm);
    const double mean_d = s / (double)n;
    double var_d = ss / (double)n - mean_d * mean_d;
    if (var_d < 0) var_d = 0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var_d + (double)eps));
    if (threadIdx.x == 0) { stats[2 * bg] = mean; stats[2 * bg + 1] = rstd; }
    if (v4) {
        const float4* x4 = reinterpret_cast<const float4*>(xr);
        float4* y4 = reinterpret_cast<float4*>(yr);
        const int64_t hw4 = HW >> 2;
        for (int64_t i = threadIdx.x; i < (n >> 2); i += blockDim.x) {
            const int64_t c = g * cg + i / hw4;
            const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
            float4 v = x4[i];
            v.x = fmaf(v.x, ga, be); v.y = fmaf(v.y, ga, be); v.z = fmaf(v.z, ga, be); v.w = fmaf(v.w, ga, be);
            if (silu) { v.x *= sigmoid_f(v.x); v.y *= sigmoid_f(v.y); v.z *= sigmoid_f(v.z); v.w *= sigmoid_f(v.w); }
            y4[i] = v;
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const int64_t c = g * cg + i / HW;
            const float ga = gamma[c] * rstd, be = beta[c] - mean * ga;
            float v = fmaf(xr[i], ga, be);
            if (silu) v *= sigmoid_f(v);
            yr[i] = v;
        }
    }
}

// dz = dy * silu'(z) (z = the normalised, affine value) when the SiLU is fused; then the GroupNorm input gradient
//   dx = rstd * (dz gamma - (S1 + xhat S2) / n),  S1 = sum dz gamma,  S2 = sum dz gamma xhat   over the group.
__global__ void __launch_bounds__(512) k_gn_bwd_nchw(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ stats, float* __restrict__ dx, int64_t C,
                                                     int64_t HW, int G, int silu) {
    __shared__ double sm[16];
    const int64_t bg = blockIdx.x;
    const int g = (int)(bg % G);
    const int64_t cg = C / G, n = cg * HW;
    const float* xr = x + bg * n;
    const float* dr = dy + bg * n;
    float* or_ = dx + bg * n;
    const float mean = stats[2 * bg], rstd = stats[2 * bg + 1];
    auto dzg = [&](float xv, float dv, float ga, float be, float& xhat) {
        xhat = (xv - mean) * rstd;
        float d = dv;
        if (silu) {
            const float z = fmaf(xhat, ga, be), sg = sigmoid_f(z);
            d *= sg * (1.0f + z * (1.0f - sg));
        }
        return d * ga;
    };
    double s1 = 0, s2 = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int64_t c = g * cg + i / HW;
        float xh;
        const float t = dzg(xr[i], dr[i], gamma[c], beta[c], xh);
        s1 += t;
        s2 += (double)t * xh;
    }
    block_sum2_d(s1, s2, sm);
    const float m1 = (float)(s1 / (double)n), m2 = (float)(s2 / (double)n);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int64_t c = g * cg + i / HW;
        float xh;
        const float t = dzg(xr[i], dr[i], gamma[c], beta[c], xh);
        or_[i] = rstd * (t - m1 - xh * m2);
    }
}

extern "C" int edadm_gn_fwd_nchw(const float* x, const float* gamma, const float* beta, float* y, float* stats, int64_t B,
                                 int64_t C, int64_t HW, int G, float eps, int silu, void* stream) {
    if (!x || !gamma || !beta || !y || !stats || B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) return EDADM_EINVAL;
    if ((HW & 3) == 0 && (((uintptr_t)x & 15) || ((uintptr_t)y & 15))) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_gn_fwd_nchw, dim3((unsigned)(B * G)), dim3(512), 0, (hipStream_t)stream, x, gamma, beta, y, stats, C, HW,
                       G, eps, silu);
    return edadm_launch_status();
}
extern "C" int edadm_gn_bwd_nchw(const float* dy, const float* x, const float* gamma, const float* beta, const float* stats,
                                 float* dx, int64_t B, int64_t C, int64_t HW, int G, int silu, void* stream) {
    if (!dy || !x || !gamma || !beta || !stats || !dx || B <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_gn_bwd_nchw, dim3((unsigned)(B * G)), dim3(512), 0, (hipStream_t)stream, dy, x, gamma, beta, stats, dx, C,
                       HW, G, silu);
    return edadm_launch_status();
}

// ------------------------------------------------------------------------------------------------ GroupNorm (+ SiLU), NHWC
// The same forward / input gradient over x [B][HW][C] (C % 4 == 0, C <= 1024): the layout the contraction kernels compute in,
// so that a reconstruction iteration of a convolutional unit runs without layout conversions between its operators (the
// NCHW form costs four conversion passes per convolution: operand in, result out, gradient in, gradient out).  A group's
// elements are strided here, so both directions are chunked reductions: per (image, chunk of rows) per-CHANNEL partial sums
// (a lane owns four channels down the chunk), a per-(image, group) reduction of the partials in fp64, then the apply pass.
//   MODE 0  partials (sum x, sum x^2)             -> stats (mean, rstd)      -> y  = [silu](x ga' + be')
//   MODE 1  partials (sum t, sum t xhat), t = dy [silu'] gamma  -> (m1, m2)  -> dx = rstd (t - m1 - xhat m2)
// ws: [B][nchunk][C][2] floats.
static int gnt_chunks(int64_t B, int64_t HW) {
    int64_t n = HW / 32, cap = 4096 / (B < 1 ? 1 : B);
    if (cap < 1) cap = 1;
    if (n > cap) n = cap;
    return (int)(n < 1 ? 1 : n);
}
struct GntCh { float mean, rstd, ga, be; };
__device__ __forceinline__ float gnt_t(float xv, float dv, const GntCh& c, int silu, float& xhat) {
    xhat = (xv - c.mean) * c.rstd;
    float d = dv;
    if (silu) {
        const float z = fmaf(xhat, c.ga, c.be), sg = sigmoid_f(z);
        d *= sg * (1.0f + z * (1.0f - sg));
    }
    return d * c.ga;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_gnt_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ stats, float* __restrict__ ws_, int64_t HW,
                                                     int64_t C, int G, int nchunk, int silu) {
    // forward statistics (MODE 0) are summed in fp64 from the first element, like k_gn_fwd_nchw: var = E[x^2] - mean^2 cancels
    // |mean| / std squared of the partials' precision, and an fp32 partial loses those digits before the fp64 reduction sees it
    using Acc = typename std::conditional<MODE == 0, double, float>::type;
    extern __shared__ double sm_raw[];
    Acc* sm = reinterpret_cast<Acc*>(sm_raw);                         // [RS][C][2]
    Acc* ws = reinterpret_cast<Acc*>(ws_);
    const int64_t b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int64_t r0 = HW * chunk / nchunk, r1 = HW * (chunk + 1) / nchunk;
    const int Q = (int)(C >> 2), RS = 256 / Q;
    const int q = threadIdx.x % Q, rs = threadIdx.x / Q;
    const float4* xa = reinterpret_cast<const float4*>(x) + b * HW * Q;
    const float4* da = reinterpret_cast<const float4*>(dy) + b * HW * Q;
    Acc s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    GntCh ch[4];
    if (MODE == 1) {
        const int cg = (int)(C / G);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = q * 4 + j, g = c / cg;
            ch[j] = GntCh{stats[(b * G + g) * 2], stats[(b * G + g) * 2 + 1], gamma[c], beta[c]};
        }
    }
    if (rs < RS) {
        for (int64_t r = r0 + rs; r < r1; r += RS) {
            const float4 v = xa[r * Q + q];
            const float e[4] = {v.x, v.y, v.z, v.w};
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { s[j] += (Acc)e[j]; ss[j] += (Acc)e[j] * (Acc)e[j]; }
            } else {
                const float4 dv = da[r * Q + q];
                const float d[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xh;
                    const float t = gnt_t(e[j], d[j], ch[j], silu, xh);
                    s[j] += t;
                    ss[j] += t * xh;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sm[((int64_t)rs * C + q * 4 + j) * 2] = s[j];
            sm[((int64_t)rs * C + q * 4 + j) * 2 + 1] = ss[j];
        }
    }
    __syncthreads();
    Acc* wb = ws + ((b * nchunk + chunk) * C) * 2;
    for (int c = threadIdx.x; c < C; c += 256) {
        Acc a = 0, bq = 0;
        for (int r = 0; r < RS; ++r) { a += sm[((int64_t)r * C + c) * 2]; bq += sm[((int64_t)r * C + c) * 2 + 1]; }
        wb[2 * c] = a;
        wb[2 * c + 1] = bq;
    }
}
template <int MODE>
__global__ void __launch_bounds__(64) k_gnt_final(const float* __restrict__ ws_, float* __restrict__ out2, int64_t HW, int64_t C,
                                                  int G, int nchunk, float eps) {
    using Acc = typename std::conditional<MODE == 0, double, float>::type;
    const Acc* ws = reinterpret_cast<const Acc*>(ws_);
    const int64_t b = blockIdx.y, g = blockIdx.x;
    const int cg = (int)(C / G);
    const int items = nchunk * cg;
    double s = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < items; i += 64) {
        const int chk = i / cg, c = (int)(g * cg) + i % cg;
        const Acc* p = ws + ((b * nchunk + chk) * C + c) * 2;
        s += (double)p[0];
        ss += (double)p[1];
    }
    s = wave_sum_d(s);
    ss = wave_sum_d(ss);
    if (threadIdx.x == 0) {
        const double n = (double)HW * cg;
        if (MODE == 0) {
            const double mean = s / n;
            double var = ss / n - mean * mean;
            if (var < 0) var = 0;
            out2[(b * G + g) * 2] = (float)mean;
            out2[(b * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
        } else {
            out2[(b * G + g) * 2] = (float)(s / n);
            out2[(b * G + g) * 2 + 1] = (float)(ss / n);
        }
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) k_gnt_apply(const float* __restrict__ x, const float* __restrict__ dy,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ stats, const float* __restrict__ m12,
                                                   float* __restrict__ out, int64_t HW, int64_t C, int G, int nchunk, int silu) {
    const int64_t b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int64_t r0 = HW * chunk / nchunk, r1 = HW * (chunk + 1) / nchunk;
    const int Q = (int)(C >> 2), RS = 256 / Q;
    const int q = threadIdx.x % Q, rs = threadIdx.x / Q;
    if (rs >= RS) return;
    const int cg = (int)(C / G);
    GntCh ch[4];
    float ga[4], be[4], m1[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = q * 4 + j, g = c / cg;
        const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
        ch[j] = GntCh{mean, rstd, gamma[c], beta[c]};
        ga[j] = gamma[c] * rstd;                                      // k_gn_fwd_nchw's arithmetic
        be[j] = beta[c] - mean * ga[j];
        if (MODE == 1) { m1[j] = m12[(b * G + g) * 2]; m2[j] = m12[(b * G + g) * 2 + 1]; }
    }
    const float4* xa = reinterpret_cast<const float4*>(x) + b * HW * Q;
    const float4* da = reinterpret_cast<const float4*>(dy) + b * HW * Q;
    float4* oa = reinterpret_cast<float4*>(out) + b * HW * Q;
    for (int64_t r = r0 + rs; r < r1; r += RS) {
        const float4 v = xa[r * Q + q];
        const float e[4] = {v.x, v.y, v.z, v.w};
        float o[4];
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = fmaf(e[j], ga[j], be[j]);
                if (silu) o[j] *= sigmoid_f(o[j]);
            }
        } else {
            const float4 dv = da[r * Q + q];
            const float d[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float xh;
                const float t = gnt_t(e[j], d[j], ch[j], silu, xh);
                o[j] = ch[j].rstd * (t - m1[j] - xh * m2[j]);
            }
        }
        oa[r * Q + q] = make_float4(o[0], o[1], o[2], o[3]);
    }
}
extern "C" int64_t edadm_gn_nhwc_ws_floats(int64_t B, int64_t HW, int64_t C, int G) {
    return B * gnt_chunks(B, HW) * C * 4 + B * G * 2;            // forward partials are doubles
}
static bool gnt_ok(int64_t B, int64_t C, int64_t HW, int G) {
    return B > 0 && B <= 65535 && C > 0 && HW > 0 && G > 0 && !(C % G) && !(C & 3) && C <= 1024;
}
extern "C" int edadm_gn_fwd_nhwc(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* ws,
                                 int64_t B, int64_t C, int64_t HW, int G, float eps, int silu, void* stream) {
    if (!x || !gamma || !beta || !y || !stats || !ws || !gnt_ok(B, C, HW, G)) return EDADM_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return EDADM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = gnt_chunks(B, HW);
    const int Q = (int)(C >> 2), RS = 256 / Q;
    const size_t smem = (size_t)RS * C * 2 * sizeof(double);
    if (((uintptr_t)ws & 7)) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_gnt_partial<0>, dim3(nchunk, (unsigned)B), dim3(256), smem, st, x, (const float*)nullptr, gamma, beta,
                       (const float*)nullptr, ws, HW, C, G, nchunk, silu);
    hipLaunchKernelGGL(k_gnt_final<0>, dim3((unsigned)G, (unsigned)B), dim3(64), 0, st, ws, stats, HW, C, G, nchunk, eps);
    hipLaunchKernelGGL(k_gnt_apply<0>, dim3(nchunk, (unsigned)B), dim3(256), 0, st, x, (const float*)nullptr, gamma, beta, stats,
                       (const float*)nullptr, y, HW, C, G, nchunk, silu);
    return edadm_launch_status();
}
extern "C" int edadm_gn_bwd_nhwc(const float* dy, const float* x, const float* gamma, const float* beta, const float* stats,
                                 float* dx, float* ws, int64_t B, int64_t C, int64_t HW, int G, int silu, void* stream) {
    if (!dy || !x || !gamma || !beta || !stats || !dx || !ws || !gnt_ok(B, C, HW, G)) return EDADM_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15)) return EDADM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = gnt_chunks(B, HW);
    const int Q = (int)(C >> 2), RS = 256 / Q;
    const size_t smem = (size_t)RS * C * 2 * sizeof(float);
    float* m12 = ws + B * nchunk * C * 2;
    hipLaunchKernelGGL(k_gnt_partial<1>, dim3(nchunk, (unsigned)B), dim3(256), smem, st, x, dy, gamma, beta, stats, ws, HW, C, G,
                       nchunk, silu);
    hipLaunchKernelGGL(k_gnt_final<1>, dim3((unsigned)G, (unsigned)B), dim3(64), 0, st, ws, m12, HW, C, G, nchunk, 0.f);
    hipLaunchKernelGGL(k_gnt_apply<1>, dim3(nchunk, (unsigned)B), dim3(256), 0, st, x, dy, gamma, beta, stats, m12, dx, HW, C, G,
                       nchunk, silu);
    return edadm_launch_status();
}

// ------------------------------------------------------------------------------------------------ LayerNorm, [rows][C]
// one wave per row, the row in registers (C <= 4096): mean, centred variance, apply -- fp32, two-pass on registers
template <int LN_MAXV>
__global__ void __launch_bounds__(256) k_ln_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float* __restrict__ y,
                                                float* __restrict__ stats, int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = j * 64 + lane;
        v[j] = c < C ? xr[c] : 0.f;
        s += v[j];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = j * 64 + lane;
        const float d = c < C ? v[j] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = j * 64 + lane;
        if (c < C) y[row * C + c] = (v[j] - mean) * rstd * gamma[c] + beta[c];
    }
}
template <int LN_MAXV>
__global__ void __launch_bounds__(256) k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                const float* __restrict__ gamma, const float* __restrict__ stats,
                                                float* __restrict__ dx, int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float t[LN_MAXV], xh[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = j * 64 + lane;
        if (c < C) {
            xh[j] = (x[row * C + c] - mean) * rstd;
            t[j] = dy[row * C + c] * gamma[c];
        } else {
            xh[j] = t[j] = 0.f;
        }
        s1 += t[j];
        s2 += t[j] * xh[j];
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int c = j * 64 + lane;
        if (c < C) dx[row * C + c] = rstd * (t[j] - m1 - xh[j] * m2);
    }
}
extern "C" int edadm_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int64_t rows,
                            int64_t C, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || !stats || rows <= 0 || C <= 0 || C > 4096) return EDADM_EINVAL;
    const dim3 grid((unsigned)((rows + 3) / 4));
#define EDADM_LN_FWD(V) hipLaunchKernelGGL(k_ln_fwd<V>, grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, stats, rows, (int)C, eps)
    if (C <= 512) EDADM_LN_FWD(8);
    else if (C <= 1536) EDADM_LN_FWD(24);
    else EDADM_LN_FWD(64);
#undef EDADM_LN_FWD
    return edadm_launch_status();
}
extern "C" int edadm_ln_bwd(const float* dy, const float* x, const float* gamma, const float* stats, float* dx, int64_t rows,
                            int64_t C, void* stream) {
    if (!dy || !x || !gamma || !stats || !dx || rows <= 0 || C <= 0 || C > 4096) return EDADM_EINVAL;
    const dim3 grid((unsigned)((rows + 3) / 4));
#define EDADM_LN_BWD(V) hipLaunchKernelGGL(k_ln_bwd<V>, grid, dim3(256), 0, (hipStream_t)stream, dy, x, gamma, stats, dx, rows, (int)C)
    if (C <= 512) EDADM_LN_BWD(8);
    else if (C <= 1536) EDADM_LN_BWD(24);
    else EDADM_LN_BWD(64);
#undef EDADM_LN_BWD
    return edadm_launch_status();
}

// ------------------------------------------------------------------------------------------------ GEGLU, [rows][2 inner]
// out = a * gelu(g) with (a, g) = the two halves of a row (attention.py:43-45); exact erf form as F.gelu
__device__ __forceinline__ float gelu_f(float g) { return 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)); }
__global__ void __launch_bounds__(256) k_geglu_fwd(const float* __restrict__ h, float* __restrict__ out, int64_t rows,
                                                   int64_t inner) {
    const int64_t n = rows * inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / inner, j = i - r * inner;
        out[i] = h[r * 2 * inner + j] * gelu_f(h[r * 2 * inner + inner + j]);
    }
}
__global__ void __launch_bounds__(256) k_geglu_bwd(const float* __restrict__ dy, const float* __restrict__ h,
                                                   float* __restrict__ dh, int64_t rows, int64_t inner) {
    const int64_t n = rows * inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / inner, j = i - r * inner;
        const float a = h[r * 2 * inner + j], g = h[r * 2 * inner + inner + j], d = dy[i];
        const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * g * g);
        dh[r * 2 * inner + j] = d * g * cdf;
        dh[r * 2 * inner + inner + j] = d * a * (cdf + g * pdf);
    }
}
extern "C" int edadm_geglu_fwd(const float* h, float* out, int64_t rows, int64_t inner, void* stream) {
    if (!h || !out || rows <= 0 || inner <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_geglu_fwd, dim3(edadm_grid(rows * inner, 256)), dim3(256), 0, (hipStream_t)stream, h, out, rows, inner);
    return edadm_launch_status();
}
extern "C" int edadm_geglu_bwd(const float* dy, const float* h, float* dh, int64_t rows, int64_t inner, void* stream) {
    if (!dy || !h || !dh || rows <= 0 || inner <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_geglu_bwd, dim3(edadm_grid(rows * inner, 256)), dim3(256), 0, (hipStream_t)stream, dy, h, dh, rows, inner);
    return edadm_launch_status();
}

// ------------------------------------------------------------------------------------------------ SiLU backward
__global__ void __launch_bounds__(256) k_silu_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                  float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i], sg = sigmoid_f(v);
        dx[i] = dy[i] * sg * (1.0f + v * (1.0f - sg));
    }
}
extern "C" int edadm_silu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
    if (!dy || !x || !dx || n <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_silu_bwd, dim3(edadm_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n);
    return edadm_launch_status();
}

// ------------------------------------------------------------------------------------------------ softmax backward
// dx = p * (dp - sum_j dp_j p_j) per row; one wave per row, any row length (strided loop)
__global__ void __launch_bounds__(256) k_softmax_bwd(const float* __restrict__ dp, const float* __restrict__ p,
                                                     float* __restrict__ dx, int64_t rows, int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = p + row * cols;
    const float* dr = dp + row * cols;
    float s = 0.f;
    for (int64_t c = lane; c < cols; c += 64) s += dr[c] * pr[c];
    s = wave_sum(s);
    for (int64_t c = lane; c < cols; c += 64) dx[row * cols + c] = pr[c] * (dr[c] - s);
}
extern "C" int edadm_softmax_bwd(const float* dp, const float* p, float* dx, int64_t rows, int64_t cols, void* stream) {
    if (!dp || !p || !dx || rows <= 0 || cols <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_softmax_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dp, p, dx, rows, cols);
    return edadm_launch_status();
}

// plain fp32 row softmax of any length (the register-resident edadm_softmax_f32 covers cols % 4 == 0, cols <= 4096)
__global__ void __launch_bounds__(256) k_softmax_fwd_any(const float* __restrict__ s, float* __restrict__ out, int64_t rows,
                                                         int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * cols;
    float mx = -INFINITY;
    for (int64_t c = lane; c < cols; c += 64) mx = fmaxf(mx, sr[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int64_t c = lane; c < cols; c += 64) sum += expf(sr[c] - mx);
    sum = wave_sum(sum);
    for (int64_t c = lane; c < cols; c += 64) out[row * cols + c] = expf(sr[c] - mx) / sum;
}
extern "C" int edadm_softmax_fwd_any(const float* s, float* out, int64_t rows, int64_t cols, void* stream) {
    if (!s || !out || rows <= 0 || cols <= 0) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_softmax_fwd_any, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, out, rows, cols);
    return edadm_launch_status();
}

// ------------------------------------------------------------------------------------------------ batched transpose
// out[z][c][r] = x[z][r][c]: 32 x 32 tiles through LDS (the K-major operands of the attention products' gradients)
__global__ void __launch_bounds__(256) k_transpose_batched(const float* __restrict__ x, float* __restrict__ out, int64_t R,
                                                           int64_t Cc) {
    __shared__ float tile[32][33];
    const int64_t z = blockIdx.z;
    const float* xs = x + z * R * Cc;
    float* os = out + z * R * Cc;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < Cc) ? xs[r * Cc + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < Cc && r < R) os[c * R + r] = tile[tx][ty + 8 * k];
    }
}
extern "C" int edadm_transpose_batched_f32(const float* x, float* out, int64_t Z, int64_t R, int64_t C, void* stream) {
    if (!x || !out || Z <= 0 || R <= 0 || C <= 0 || Z > 65535) return EDADM_EINVAL;
    hipLaunchKernelGGL(k_transpose_batched, dim3((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32), (unsigned)Z), dim3(256), 0,
                       (hipStream_t)stream, x, out, R, C);
    return edadm_launch_status();
}
