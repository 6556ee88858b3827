// fp32 3x3 convolution for the network's last layer (N = 3 or 4 output channels): its activation
// quantizer is disabled (quant_model.py:90-95) so the operand stays fp32; 2*9*C*N flops per pixel is
// <0.1 % of a UNet forward.  A wave keeps the whole filter in registers (lane l owns channels l, l+64, ...:
// N*9*ceil(C/64) floats) and streams pixels: per tap one coalesced 256-byte read per 64 channels, N FMAs,
// and one butterfly reduction per output.
#include "common.h"
#include "../../include/edadm.h"

#define SMALLN_MAX 4
#define SMALLN_P 4          // output pixels per strip: a row of the input is loaded once for its three horizontal taps
// One block per output row (b, y), a wave per 16-pixel span in strips of SMALLN_P pixels.  A lane owns channels
// lane + 64 j; the 9 x N weight vectors of those channels stay in registers; per strip the 3 x (P + 2) input
// pixels are read once (coalesced 256-byte rows) and feed every tap that uses them.
template <int CJ>
__global__ void __launch_bounds__(256) k_conv3x3_smalln(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int64_t B, int64_t H, int64_t W, int64_t C, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float wr[SMALLN_MAX][9][CJ];
#pragma unroll
    for (int n = 0; n < SMALLN_MAX; ++n)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int64_t c = lane + 64 * j;
                wr[n][t][j] = (n < N && c < C) ? w[((int64_t)n * 9 + t) * C + c] : 0.f;
            }
    const int iW = (int)W, iH = (int)H;
    const int nstrips = (iW + SMALLN_P - 1) / SMALLN_P;
    // a block walks output rows (b, y) with a grid stride: the 9 x N weight vectors are loaded once per block
    for (unsigned by = blockIdx.x; by < (unsigned)(B * H); by += gridDim.x) {
    const int b = (int)(by / (unsigned)iH), y = (int)(by - (unsigned)b * (unsigned)iH);
    const float* xb = x + (int64_t)b * H * W * C;
    for (int sidx = wave; sidx < nstrips; sidx += 4) {
        const int x0 = sidx * SMALLN_P;
        float acc[SMALLN_P][SMALLN_MAX];
#pragma unroll
        for (int p = 0; p < SMALLN_P; ++p)
#pragma unroll
            for (int n = 0; n < SMALLN_MAX; ++n) acc[p][n] = 0.f;
        // all 3 x (P + 2) x CJ input values of the strip are requested before the first FMA (clamped addresses, the
        // padding zeroed afterwards): with a branch per tap the loads of each tap waited out their latency alone
        float in[3][SMALLN_P + 2][CJ];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            const int cy = iy < 0 ? 0 : (iy >= iH ? iH - 1 : iy);
            const float* xrow = xb + (int64_t)cy * W * C;
#pragma unroll
            for (int xx = -1; xx <= SMALLN_P; ++xx) {
                const int ix = x0 + xx;
                const int cx = ix < 0 ? 0 : (ix >= iW ? iW - 1 : ix);
#pragma unroll
                for (int j = 0; j < CJ; ++j) {
                    const int c = lane + 64 * j;
                    in[ky][xx + 1][j] = c < C ? xrow[(int64_t)cx * C + c] : 0.f;
                }
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            const bool yok = iy >= 0 && iy < iH;
#pragma unroll
            for (int xx = -1; xx <= SMALLN_P; ++xx) {
                const int ix = x0 + xx;
                const bool ok = yok && ix >= 0 && ix < iW;                 // wave-uniform
                float v[CJ];
#pragma unroll
                for (int j = 0; j < CJ; ++j) v[j] = ok ? in[ky][xx + 1][j] : 0.f;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int p = xx - kx + 1;                         // output pixel this (input, tap) pair feeds
                    if (p < 0 || p >= SMALLN_P) continue;
#pragma unroll
                    for (int j = 0; j < CJ; ++j)
#pragma unroll
                        for (int n = 0; n < SMALLN_MAX; ++n) acc[p][n] += v[j] * wr[n][ky * 3 + kx][j];
                }
            }
        }
        // 16 sums (4 pixels x 4 outputs) over the 64 lanes: a halving butterfly -- at each of the four upper lane bits a
        // lane keeps half of its values and hands the other half to its partner -- needs 8+4+2+1+2 shuffles instead of
        // 16 x 6 (the shuffles go through the LDS crossbar and were most of this kernel's time)
        float v16[16];
#pragma unroll
        for (int p = 0; p < SMALLN_P; ++p)
#pragma unroll
            for (int n = 0; n < SMALLN_MAX; ++n) v16[p * 4 + n] = acc[p][n];
#pragma unroll
        for (int bit = 5, cnt = 8; bit >= 2; --bit, cnt >>= 1) {
            const bool up = (lane >> bit) & 1;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < cnt) {
                    const float lo = v16[k], hi = v16[k + cnt];
                    const float keep = up ? hi : lo, send = up ? lo : hi;
                    v16[k] = keep + __shfl_xor(send, 1 << bit, 64);
                }
            }
        }
        float tot = v16[0];
        tot += __shfl_xor(tot, 2, 64);
        tot += __shfl_xor(tot, 1, 64);
        const int idx = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
        const int p = idx >> 2, n = idx & 3;
        if ((lane & 3) == 0 && n < N && x0 + p < iW)
            out[(((int64_t)b * H + y) * W + x0 + p) * N + n] = tot + (bias ? bias[n] : 0.f);
    }
    }
}
extern "C" int edadm_conv3x3_f32_smalln(const float* x, const float* w, const float* bias, float* out, int64_t B,
                                        int64_t H, int64_t W, int64_t C, int64_t N, void* stream) {
    if (!x || !w || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 320 || N <= 0 || N > SMALLN_MAX ||
        B * H > 0x7fffffff)
        return EDADM_EINVAL;
    const int64_t g = B * H < 1536 ? B * H : 1536;
    const int cj = (int)((C + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
#define SMALLN_CASE(CJ_)                                                                                          \
    if (cj == CJ_) {                                                                                              \
        hipLaunchKernelGGL(k_conv3x3_smalln<CJ_>, dim3((unsigned)g), dim3(256), 0, st, x, w, bias, out, B, H, W, C, \
                           (int)N);                                                                               \
        return edadm_launch_status();                                                                             \
    }
    SMALLN_CASE(1) SMALLN_CASE(2) SMALLN_CASE(3) SMALLN_CASE(4) SMALLN_CASE(5)
#undef SMALLN_CASE
    return EDADM_EINVAL;
}

// ---- codebook lookup of the VQ first stage: out[r] = codebook[argmin_j |z_r - e_j|^2]  (VQModelInterface.decode -> self.quantize,
// ldm/models/autoencoder.py:274-277; the quantiser itself is taming's VectorQuantizer2, not vendored by the reference: the rule here is
// its published one -- d = |z|^2 + |e|^2 - 2 z.e, first minimum -- parity unpinned).  D <= 8 floats per code.  The distance matrix
// (65536 x 8192 floats for a 16-image chunk of VQ-f4) is never formed: a workgroup stages the codebook through LDS in chunks of 2048
// codes (+ their squared norms), a thread owns one latent vector and walks the chunk with broadcast reads.
#define VQ_CHUNK 2048
template <int D>
__global__ void __launch_bounds__(256) k_vq_nearest(const float* __restrict__ z, const float* __restrict__ cb, float* __restrict__ out,
                                                    int64_t* __restrict__ idx, int64_t R, int E) {
    __shared__ float sc[VQ_CHUNK * D];
    __shared__ float s2[VQ_CHUNK];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float v[D];
    float z2 = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        v[d] = r < R ? z[r * D + d] : 0.f;
        z2 += v[d] * v[d];
    }
    float best = INFINITY;
    int bi = 0;
    for (int e0 = 0; e0 < E; e0 += VQ_CHUNK) {
        const int n = E - e0 < VQ_CHUNK ? E - e0 : VQ_CHUNK;
        __syncthreads();
        for (int i = threadIdx.x; i < n * D; i += 256) sc[i] = cb[(int64_t)e0 * D + i];
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += 256) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) a += sc[j * D + d] * sc[j * D + d];
            s2[j] = a;
        }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) dot += v[d] * sc[j * D + d];
            const float dist = (z2 - 2.0f * dot) + s2[j];
            if (dist < best) { best = dist; bi = e0 + j; }
        }
    }
    if (r < R) {
#pragma unroll
        for (int d = 0; d < D; ++d) out[r * D + d] = cb[(int64_t)bi * D + d];
        if (idx) idx[r] = bi;
    }
}
extern "C" int edadm_vq_nearest(const float* z, const float* codebook, float* out, int64_t* idx, int64_t R, int64_t D, int64_t E,
                                void* stream) {
    if (!z || !codebook || !out || R <= 0 || D <= 0 || D > 8 || E <= 0 || E > 0x7fffffff) return EDADM_EINVAL;
    const dim3 grid((unsigned)((R + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
#define VQ_CASE(D_)                                                                                            \
    if (D == D_) {                                                                                             \
        hipLaunchKernelGGL(k_vq_nearest<D_>, grid, dim3(256), 0, st, z, codebook, out, idx, R, (int)E);        \
        return edadm_launch_status();                                                                          \
    }
    VQ_CASE(1) VQ_CASE(2) VQ_CASE(3) VQ_CASE(4) VQ_CASE(5) VQ_CASE(6) VQ_CASE(7) VQ_CASE(8)
#undef VQ_CASE
    return EDADM_EINVAL;
}

// ---- TDAC step scores (scripts/calibration.py:47-69 of the reference): for every pair (i, j) of the T feature maps F[t] = [B][C][P]
// (the mid-block attention input of sampling step t, P = H W positions) the mean squared difference -- the density test
// `mean((F_i - F_j)^2) <= r` -- and the variety term sum over (b, p) of 1 - cos(F_i[b, :, p], F_j[b, :, p]), the cosine along the
// channel axis with each norm clamped at eps (torch.nn.functional.cosine_similarity).  The reference runs T (T - 1) = 380 pairs x
// five torch passes; here one workgroup per unordered pair reads the two maps once: a thread owns positions (b, p) and walks the
// channels (coalesced over p), the per-thread partials are reduced in a FIXED order (wave butterfly, then waves 0..3): deterministic.
// Both outputs are [T][T] (symmetric, zero diagonal); the host counts / adds them in the reference's order of j.
__global__ void __launch_bounds__(256) k_tdac_pairs(const float* __restrict__ F, int T, int64_t B, int64_t C, int64_t P, float eps,
                                                    float* __restrict__ mse, float* __restrict__ cosd) {
    // unordered pair index -> (i, j), i < j
    int i = 0, rem = (int)blockIdx.x;
    while (rem >= T - 1 - i) { rem -= T - 1 - i; ++i; }
    const int j = i + 1 + rem;
    const int64_t per = B * C * P;
    const float* a = F + (int64_t)i * per;
    const float* b = F + (int64_t)j * per;
    double sq = 0.0, cd = 0.0;
    for (int64_t q = threadIdx.x; q < B * P; q += 256) {
        const int64_t bb = q / P, p = q - bb * P;
        const float* pa = a + bb * C * P + p;
        const float* pb = b + bb * C * P + p;
        float dot = 0.f, na = 0.f, nb = 0.f, s2 = 0.f;
        for (int64_t c = 0; c < C; ++c) {
            const float x = pa[c * P], y = pb[c * P];
            dot = fmaf(x, y, dot);
            na = fmaf(x, x, na);
            nb = fmaf(y, y, nb);
            const float d = x - y;
            s2 = fmaf(d, d, s2);
        }
        sq += (double)s2;
        cd += (double)(1.0f - dot / (fmaxf(sqrtf(na), eps) * fmaxf(sqrtf(nb), eps)));
    }
    __shared__ double sm[2][4];
    sq = wave_sum_d(sq);
    cd = wave_sum_d(cd);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = sq; sm[1][threadIdx.x >> 6] = cd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = ((sm[0][0] + sm[0][1]) + sm[0][2]) + sm[0][3], c2 = ((sm[1][0] + sm[1][1]) + sm[1][2]) + sm[1][3];
        const float m = (float)(s / (double)per), cv = (float)c2;
        mse[(int64_t)i * T + j] = mse[(int64_t)j * T + i] = m;
        cosd[(int64_t)i * T + j] = cosd[(int64_t)j * T + i] = cv;
    }
    if (blockIdx.x == 0)                                        // the diagonal: any T (Church samples 500 steps by default)
        for (int64_t d = threadIdx.x; d < T; d += 256) mse[d * T + d] = cosd[d * T + d] = 0.f;
}
extern "C" int edadm_tdac_pair_scores(const float* feats, int64_t T, int64_t B, int64_t C, int64_t P, float eps, float* mse,
                                      float* cosdis, void* stream) {
    if (!feats || !mse || !cosdis || T < 2 || T > 32768 || B <= 0 || C <= 0 || P <= 0) return EDADM_EINVAL;
    const unsigned pairs = (unsigned)(T * (T - 1) / 2);
    hipLaunchKernelGGL(k_tdac_pairs, dim3(pairs), dim3(256), 0, (hipStream_t)stream, feats, (int)T, B, C, P, eps, mse, cosdis);
    return edadm_launch_status();
}
