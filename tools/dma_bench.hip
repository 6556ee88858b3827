// Diagnostic: LDS-DMA (global_load_lds_dwordx4) intake per CU as a function of loader waves, steps in flight,
// row width and source footprint.  One workgroup per CU, no consumers.  hipcc --offload-arch=gfx950 -O3 dma_bench.hip -o dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// variant 1: set M0, no save/restore (clobber accepted with a warning)
__device__ __forceinline__ void glds16_set(const void* gsrc, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_addr) : "memory");
}
// variant 2: four pieces behind ONE M0 write; the immediate offset moves the LDS destination (and the global address,
// compensated in the pointer)
__device__ __forceinline__ void glds16_x4(const uint8_t* g0, const uint8_t* g1, const uint8_t* g2, const uint8_t* g3, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %3, off offset:3072"
                 : : "v"(g0), "v"(g1 - 1024), "v"(g2 - 2048), "v"(g3 - 3072), "s"(lds_addr) : "memory");
}

template <int P, int D, int RB, int VAR>   // pieces per wave per step, steps in flight, row bytes (64/128)
__global__ void __launch_bounds__(1024) k_dma(const uint8_t* __restrict__ src, int64_t ld, int64_t rows_total, int iters,
                                              unsigned long long* __restrict__ ticks, unsigned* check) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    constexpr int CPR = RB / 16, RPP = 64 / CPR;
    const int prow = lane / CPR, pch = lane % CPR;
    // this wave's slice of the ring: D+1 slots of P KiB
    const uint32_t ring = lds0 + wave * (D + 1) * P * 1024;
    const unsigned rmask = (unsigned)rows_total - 1;   // rows_total is a power of two
    unsigned row = ((unsigned)blockIdx.x * nw + wave) * 977u & rmask;     // scattered start
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (VAR == 3) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(1))) u32x4* gp4;
            u32x4 v[P];
#pragma unroll
            for (int p = 0; p < P; ++p) v[p] = *(gp4)(uintptr_t)(src + ((int64_t)((row + p * RPP + prow) & rmask)) * ld + pch * 16);
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(smem + (ring - lds0) + (slot * P + p) * 1024 + lane * 16) = v[p];
        } else if constexpr (VAR == 2) {
            static_assert(VAR != 2 || P % 4 == 0, "x4 variant");
#pragma unroll
            for (int p = 0; p < P; p += 4) {
                const uint8_t* gp[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) gp[q] = src + (int64_t)((row + (p + q) * RPP + prow) & rmask) * ld + pch * 16;
                glds16_x4(gp[0], gp[1], gp[2], gp[3], ring + (uint32_t)((slot * P + p) * 1024));
            }
        } else {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int64_t r = (int64_t)((row + p * RPP + prow) & rmask);
                if constexpr (VAR == 1) glds16_set(src + r * ld + pch * 16, ring + (uint32_t)((slot * P + p) * 1024));
                else glds16(src + r * ld + pch * 16, ring + (uint32_t)((slot * P + p) * 1024));
            }
        }
        if (check && it == iters - 1) {      // verify the last step's bytes
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned bad = 0;
            for (int p = 0; p < P; ++p) {
                const int64_t r = (int64_t)((row + p * RPP + prow) & rmask);
                const uint4 want = *reinterpret_cast<const uint4*>(src + r * ld + pch * 16);
                const uint4 got = *reinterpret_cast<const uint4*>(smem + (ring - lds0) + (slot * P + p) * 1024 + lane * 16);
                bad += (want.x != got.x) | (want.y != got.y) | (want.z != got.z) | (want.w != got.w);
            }
            if (bad) atomicAdd(check, bad);
        }
        row = (row + P * RPP * 131) & rmask;
        slot = slot == D ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * P) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int P, int D, int RB, int VAR>
static void run(int nw, int64_t footprint_mb, int iters) {
    const int64_t ld = 256;
    const int64_t rows = footprint_mb * 1024 * 1024 / ld;
    uint8_t* src;
    unsigned long long* ticks;
    hipMalloc(&src, rows * ld);
    { std::vector<uint8_t> h(rows * ld); for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)((i * 2654435761u) >> 13); hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice); }
    unsigned* check; hipMalloc(&check, 4); hipMemset(check, 0, 4);
    hipMalloc(&ticks, 256 * 8);
    const size_t lds = (size_t)nw * (D + 1) * P * 1024;
    hipFuncSetAttribute((const void*)k_dma<P, D, RB, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_dma<P, D, RB, VAR>), dim3(256), dim3(nw * 64), lds, 0, src, ld, rows, iters, ticks, check);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= 256;
    const double bytes_cu = (double)nw * iters * P * 1024;
    unsigned hb = 0; hipMemcpy(&hb, check, 4, hipMemcpyDeviceToHost);
    printf("var=%d bad=%u waves=%2d P=%d D=%d row=%3dB footprint=%5lld MB : %.1f B/tick/CU  %.1f GB/s/CU  chip %.2f TB/s  (%.0f ticks/piece/wave)\n", VAR, hb, nw, P, D, RB,
           (long long)footprint_mb, bytes_cu / avg, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * 256 / (ms * 1e-3) / 1e12, avg / (iters * P));
    hipFree(src); hipFree(ticks);
}

int main() {
    for (int64_t fp : {1, 2, 8, 64}) {
        for (int nw : {1, 4, 8}) {
            run<4, 2, 128, 0>(nw, fp, 2000);
            run<4, 2, 128, 3>(nw, fp, 2000);
        }
        run<8, 2, 64, 0>(4, fp, 2000);
        run<8, 2, 64, 3>(4, fp, 2000);
    }
    return 0;
}
