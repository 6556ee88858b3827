// fp32 3x3 convolution for the network's last layer (N = 3 or 4 output channels): its activation
// quantizer is disabled (quant_model.py:90-95) so the operand stays fp32; 2*9*C*N flops per pixel is
// <0.1 % of a UNet forward.  One wave per output pixel: lanes stride the C channels of each tap
// (coalesced 256-byte reads), N partial dot products per lane, wave reduction.
#include "common.h"
#include "../../include/edadm.h"

#define SMALLN_MAX 8
__global__ void __launch_bounds__(256) k_conv3x3_smalln(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int64_t B, int64_t H, int64_t W, int64_t C, int N) {
    const int lane = threadIdx.x & 63;
    const int64_t npix = B * H * W;
    for (int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (int64_t)gridDim.x * 4) {
        const int64_t b = pix / (H * W), r = pix - b * H * W, y = r / W, xx = r - y * W;
        float acc[SMALLN_MAX];
#pragma unroll
        for (int n = 0; n < SMALLN_MAX; ++n) acc[n] = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int64_t iy = y + tap / 3 - 1, ix = xx + tap % 3 - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const float* xp = x + ((b * H + iy) * W + ix) * C;
            for (int64_t c = lane; c < C; c += 64) {
                const float v = xp[c];
#pragma unroll
                for (int n = 0; n < SMALLN_MAX; ++n)
                    if (n < N) acc[n] += v * w[((int64_t)n * 9 + tap) * C + c];
            }
        }
#pragma unroll
        for (int n = 0; n < SMALLN_MAX; ++n) {
            if (n < N) {
                const float s = wave_sum(acc[n]);
                if (lane == 0) out[pix * N + n] = s + (bias ? bias[n] : 0.f);
            }
        }
    }
}
extern "C" int edadm_conv3x3_f32_smalln(const float* x, const float* w, const float* bias, float* out, int64_t B,
                                        int64_t H, int64_t W, int64_t C, int64_t N, void* stream) {
    if (!x || !w || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || N > SMALLN_MAX) return EDADM_EINVAL;
    int64_t g = (B * H * W + 3) / 4;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_conv3x3_smalln, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, B, H,
                       W, C, (int)N);
    return edadm_launch_status();
}
