// Diagnostic: issue rate of v_mfma_i32_32x32x32_i8 on gfx950 -- one and two waves per SIMD, bare and with the ds_read_b128
// stream of k_conv3_direct's main loop (5 fragment reads per 6 MFMAs).  hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, uint32_t lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_off) : "memory");
}
#define MM(x, y, z) z = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, z, 0, 0, 0)
// MODE 0 bare; 1 read-then-compute; 2 fragments of group g + 1 requested before the MFMAs of group g; 3 = 2 + s_barrier every
// 6 groups (a step of k_conv3_direct); 4 = 3 + five 1 KiB LDS-DMA pieces per wave and step (weights, L2-resident source)
template <int MODE>
__global__ void __launch_bounds__(512) k(int iters, unsigned long long* out, int* sink, const uint8_t* wsrc) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[144 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 144 * 1024 / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    v16i acc[6];
    for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t* base = lds + (wave & 3) * 16384;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const int fr = lane & 31, fh = lane >> 5;
    v4i fa[2][2], fb[2][3];
    auto load = [&](int g, int set) {
        const int o = (g & 7) * 2048;
        fa[set][0] = *reinterpret_cast<const v4i*>(base + o + fr * 64 + (((2 * 0 + fh) ^ ((fr >> 2) & 3)) << 4));
        fa[set][1] = *reinterpret_cast<const v4i*>(base + o + (fr + 32) * 64 + (((2 * 0 + fh) ^ (((fr + 32) >> 2) & 3)) << 4));
        fb[set][0] = *reinterpret_cast<const v4i*>(base + 8192 + o + fr * 64 + (((2 * 1 + fh) ^ ((fr >> 2) & 3)) << 4));
        fb[set][1] = *reinterpret_cast<const v4i*>(base + 8192 + o + (fr + 32) * 64 + (((2 * 1 + fh) ^ (((fr + 32) >> 2) & 3)) << 4));
        fb[set][2] = *reinterpret_cast<const v4i*>(base + 4096 + o + fr * 64 + (((2 * 1 + fh) ^ ((fr >> 2) & 3)) << 4));
    };
    for (int s2 = 0; s2 < 2; ++s2) { fa[s2][0] = fa[s2][1] = v4i{lane, 1, 2, 3}; fb[s2][0] = fb[s2][1] = fb[s2][2] = v4i{4, lane, 6, 7}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {                     // one iteration = one step = 6 groups of 6 MFMAs
        if (MODE >= 3) {
            if (MODE >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (MODE >= 4) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    int q = wave + 8 * i; if (q > 35) q = 35;
                    glds16(wsrc + ((it & 15) * 36 + q) * 1024 + lane * 16, lds0 + 65536u + (uint32_t)(it & 1) * 36864u + q * 1024u);
                }
            }
        }
        if (MODE >= 2) load(0, 0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (MODE == 1) load(g, g & 1);
            if (MODE >= 2 && g + 1 < 6) load(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int t = g & 1;
            MM(fa[t][0], fb[t][0], acc[0]); MM(fa[t][0], fb[t][1], acc[1]); MM(fa[t][0], fb[t][2], acc[2]);
            MM(fa[t][1], fb[t][0], acc[3]); MM(fa[t][1], fb[t][1], acc[4]); MM(fa[t][1], fb[t][2], acc[5]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    int s = 0;
    for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (s == 0x7fffffff) sink[0] = s;
}

int main() {
    unsigned long long* d; int* sink; uint8_t* wsrc;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 4); hipMalloc(&wsrc, 16 * 36 * 1024); hipMemset(wsrc, 1, 16 * 36 * 1024);
    unsigned long long h[256 * 8];
    const int iters = 400;
    const char* names[5] = {"bare", "read, then compute", "reads one group ahead", "+ barrier per step", "+ 5 DMA pieces per wave and step"};
    for (int mode = 0; mode < 5; ++mode)
        for (int waves = 4; waves <= 8; waves += 4) {
            if (mode >= 3 && waves == 4) continue;
            const int grid = 256;
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64 * waves), 0, 0, iters, d, sink, wsrc); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64 * waves), 0, 0, iters, d, sink, wsrc); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64 * waves), 0, 0, iters, d, sink, wsrc); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64 * waves), 0, 0, iters, d, sink, wsrc); break;
                    default: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64 * waves), 0, 0, iters, d, sink, wsrc); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double mx = 0, mn = 1e30;
            for (int w = 0; w < waves; ++w) { mx = h[w] > mx ? h[w] : mx; mn = h[w] < mn ? h[w] : mn; }
            printf("mode %d (%s) waves/CU %d: cycles per step (36 MFMAs) per wave: min %.0f max %.0f -> per MFMA and SIMD %.1f\n", mode,
                   names[mode], waves, mn / iters, mx / iters, mx / (36.0 * iters) / (waves / 4));
        }
    return 0;
}
