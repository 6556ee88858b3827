// fp32 3x3 convolution for the network's last layer (N = 3 or 4 output channels): its activation
// quantizer is disabled (quant_model.py:90-95) so the operand stays fp32; 2*9*C*N flops per pixel is
// <0.1 % of a UNet forward.  A wave keeps the whole filter in registers (lane l owns channels l, l+64, ...:
// N*9*ceil(C/64) floats) and streams pixels: per tap one coalesced 256-byte read per 64 channels, N FMAs,
// and one butterfly reduction per output.
#include "common.h"
#include "../../include/edadm.h"

#define SMALLN_MAX 4
template <int CJ>
__global__ void __launch_bounds__(256) k_conv3x3_smalln(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int64_t B, int64_t H, int64_t W, int64_t C, int N) {
    const int lane = threadIdx.x & 63;
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
    const int64_t npix = B * H * W;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t pix = wave0; pix < npix; pix += nwaves) {
        const int64_t b = pix / (H * W), r = pix - b * H * W, y = r / W, xx = r - y * W;
        float acc[SMALLN_MAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int64_t iy = y + t / 3 - 1, ix = xx + t % 3 - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;      // wave-uniform
            const float* xp = x + ((b * H + iy) * W + ix) * C;
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                const int64_t c = lane + 64 * j;
                const float v = c < C ? xp[c] : 0.f;
#pragma unroll
                for (int n = 0; n < SMALLN_MAX; ++n) acc[n] += v * wr[n][t][j];
            }
        }
#pragma unroll
        for (int n = 0; n < SMALLN_MAX; ++n) {
            const float s = wave_sum(acc[n]);
            if (lane == 0 && n < N) out[pix * N + n] = s + (bias ? bias[n] : 0.f);
        }
    }
}
extern "C" int edadm_conv3x3_f32_smalln(const float* x, const float* w, const float* bias, float* out, int64_t B,
                                        int64_t H, int64_t W, int64_t C, int64_t N, void* stream) {
    if (!x || !w || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 320 || N <= 0 || N > SMALLN_MAX)
        return EDADM_EINVAL;
    int64_t g = (B * H * W + 3) / 4;
    if (g > 2048) g = 2048;
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
