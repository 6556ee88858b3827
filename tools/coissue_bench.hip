// Diagnostic: do the matrix pipe and the vector ALU of a SIMD run side by side?  One 512-thread workgroup per CU = two waves per SIMD:
// waves 0-3 issue NM back-to-back int8 MFMAs (independent accumulators), waves 4-7 issue NV fp32 FMAs (four independent chains).
// Timed alone and together (HIP events over a grid of one workgroup per CU).  hipcc --offload-arch=gfx950 -O3 tools/coissue_bench.hip -o tools/coissue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(float* out, int nm, int nv) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        v16i a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        v4i x = {(int)threadIdx.x, 1, 2, 3}, y = {3, 2, 1, (int)threadIdx.x};
        for (int i = 0; i < nm; i += 4) {
            a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a3, 0, 0, 0);
        }
        if (nm < 0) out[threadIdx.x] = (float)(a0[0] + a1[1] + a2[2] + a3[3]);
        if (nm > 0 && a0[0] == 0x7fffffff) out[threadIdx.x] = (float)(a0[1] + a1[1] + a2[2] + a3[3]);
    } else {
        float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f;
        for (int i = 0; i < nv; i += 4) {
            f0 = __builtin_fmaf(f0, 1.0001f, 0.5f);
            f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
            f2 = __builtin_fmaf(f2, 1.0001f, 0.5f);
            f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
            asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
        }
        if (f0 + f1 + f2 + f3 == 12345.678f) out[threadIdx.x] = f0;
    }
}

static float run(float* out, int nm, int nv) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, nm, nv);
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, nm, nv);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / 5 * 1e3f;
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4096);
    const int NM = 20000;                                   // 20000 MFMAs x 32 cycles = 640 k cycles per wave
    for (int nv : {40000, 80000, 160000, 320000}) {         // x 4 cycles
        const float tm = run(out, NM, 0), tv = run(out, 0, nv), tb = run(out, NM, nv);
        printf("MFMA alone %8.1f us | %6d FMAs alone %8.1f us | together %8.1f us  (sum %8.1f, max %8.1f)\n", tm, nv, tv, tb, tm + tv, tm > tv ? tm : tv);
    }
    return 0;
}
