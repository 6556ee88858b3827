// Diagnostic: fp32 tile epilogue store patterns on gfx950 -- (A) the product's: a lane owns one column of a 32-wide block, a store
// instruction writes two full 128-byte row segments (4 B per lane, 96 instructions per 64 x 96 wave tile); (B) 16 B per lane, eight
// full 128-byte row segments per instruction (24 instructions per wave tile).  With / without a residual read in the same pattern.
// hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o tools/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int PAT, bool RES>
__global__ void __launch_bounds__(512) k_store(float* __restrict__ out, const float* __restrict__ res, int64_t ldo, int spin) {
    __shared__ float pad[36 * 1024];                         // ~144 KB: one workgroup per CU, like the convolution
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t row0 = (int64_t)blockIdx.x * 256 + wm * 64, col0 = wn * 96;
    if (spin < 0) pad[threadIdx.x] = 0.f;
    float v = (float)lane;
    // a main-loop stand-in so that the workgroups do not all store at once for nothing: `spin` dependent FMAs
    for (int s = 0; s < spin; ++s) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    if constexpr (PAT == 0) {
        const int fr = lane & 31, fh4 = (lane >> 5) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int64_t r = row0 + i * 32 + 8 * g + e + fh4;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        float x = v + (float)(i + g + e + j);
                        if constexpr (RES) x += __builtin_nontemporal_load(res + r * ldo + col0 + j * 32 + fr);
                        __builtin_nontemporal_store(x, out + r * ldo + col0 + j * 32 + fr);
                    }
                }
    } else {
        const int rr = lane >> 3, cq = (lane & 7) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int64_t r = row0 + i * 32 + 8 * g + rr;
                    f4 x = {v + i, v + g, v + j, v};
                    f4* p = reinterpret_cast<f4*>(out + r * ldo + col0 + j * 32 + cq);
                    if constexpr (RES) x += __builtin_nontemporal_load(reinterpret_cast<const f4*>(res + r * ldo + col0 + j * 32 + cq));
                    __builtin_nontemporal_store(x, p);
                }
    }
}

template <int PAT, bool RES>
static float run(float* out, const float* res, int tiles, int spin, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_store<PAT, RES>), dim3(tiles), dim3(512), 0, 0, out, res, 192, spin);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_store<PAT, RES>), dim3(tiles), dim3(512), 0, 0, out, res, 192, spin);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f;
}

int main() {
    const int tiles = 1600;                                  // 409600 x 192 fp32
    const size_t n = (size_t)tiles * 256 * 192;
    float *out, *res;
    hipMalloc(&out, n * 4); hipMalloc(&res, n * 4);
    hipMemset(res, 0, n * 4);
    for (int spin : {0, 4000, 12000}) {
        const float a0 = run<0, false>(out, res, tiles, spin, 10), b0 = run<1, false>(out, res, tiles, spin, 10);
        const float a1 = run<0, true>(out, res, tiles, spin, 10), b1 = run<1, true>(out, res, tiles, spin, 10);
        printf("spin %5d | store only: 4 B/lane %7.1f us (%.2f TB/s)  16 B/lane %7.1f us (%.2f TB/s) | with residual: 4 B/lane %7.1f us (%.2f TB/s)  16 B/lane %7.1f us (%.2f TB/s)\n",
               spin, a0, n * 4 / a0 / 1e6, b0, n * 4 / b0 / 1e6, a1, n * 8 / a1 / 1e6, b1, n * 8 / b1 / 1e6);
    }
    return 0;
}
