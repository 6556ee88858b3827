// K4 / K6 contractions on the gfx950 matrix cores.
//
// One kernel template serves the quantised conv / linear layers (int8 operands,
// v_mfma_i32_32x32x32_i8, implicit-GEMM gather over NHWC activations) and the attention
// products (integer-valued f16 operands, v_mfma_f32_32x32x16_f16; exact in fp32).
//
// Tile: 256 threads = 4 waves as 2(M) x 2(N); each wave owns TM x TN MFMA tiles of 32x32, so a
// workgroup computes (64*TM) x (64*TN) outputs.  K advances 64 BYTES per step (64 int8 / 32 f16):
// every row of a tile is one 64-byte segment = four 16-byte chunks; chunk c of row r lives at
// LDS chunk c ^ ((r >> 2) & 3), which makes the ds_read_b128 fragment reads (lane -> row lane&31,
// chunk 2*ks + (lane>>5)) hit 16 distinct 16-byte slots per 16-lane group (conflict-free).
// Global -> register -> LDS staging with the next tile's loads issued before the current tile's
// MFMAs (double-buffered LDS, one barrier per K-step).  A and B fragments use the same
// (lane, byte) -> k map, so the dot product is independent of the hardware's k numbering.
#include "common.h"
#include "../../include/edadm.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>
#ifndef EDADM_USE_NT8
#define EDADM_USE_NT8 1
#endif
#ifndef EDADM_GEMM_STAGES
#define EDADM_GEMM_STAGES 3
#endif

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// Direct-to-LDS 16-byte load issued from inline asm: hipcc's waitcnt pass would otherwise put
// `s_waitcnt vmcnt(0)` in front of every ds_read that may alias an in-flight LDS-DMA write, which
// serialises the pipeline (seen in the .s).  M0 carries the wave-uniform LDS byte address and is
// restored inside the same statement; completion is tracked by the hand-counted vmcnt below.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

struct ConvGeom {
    int mode, B, H, W, Cin, Ho, Wo, KH, KW, stride, pad0, ups, padval, r0, r1, r2;
};

// operand types: 0 = int8 (i32 accumulate), 1 = f16, 2 = f32 (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain; the
// calibration graph's contraction).  A 16-byte fragment is 16 / 8 / 4 k-values.
template <int DT>
struct Acc { typedef v16f type; };
template <>
struct Acc<0> { typedef v16i type; };

template <int DT>
__device__ __forceinline__ void mma_step(const uint4& fa, const uint4& fb, typename Acc<DT>::type& acc) {
    if constexpr (DT == 0) {
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const v4i*>(&fa), *reinterpret_cast<const v4i*>(&fb),
                                                    acc, 0, 0, 0);
    } else if constexpr (DT == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const v8h*>(&fa), *reinterpret_cast<const v8h*>(&fb),
                                                     acc, 0, 0, 0);
    } else {
        const float* a = reinterpret_cast<const float*>(&fa);
        const float* b = reinterpret_cast<const float*>(&fb);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
}

// round(v / d) with the reference's true-division result at the cost of a multiply: t = v * (1/d) differs from
// v / d by a couple of ulp, which can only change the rounded integer when t sits within 1e-3 of a .5
// boundary; those (rare) lanes redo the IEEE division.
__device__ __forceinline__ float rint_div(float v, float d, float inv_d) {
    float t = v * inv_d;
    const float f = t - floorf(t);
    if (fabsf(f - 0.5f) < 1e-3f) t = v / d;
    return rintf(t);
}

// Per-column epilogue constants live in LDS (scale, bias, and the time-embedding rows of the few batch
// entries a tile spans), so the store phase issues no global load except the residual, and all residual
// loads of a slab are issued before its first store: on gfx950 vmcnt counts stores too and retires in
// order, so a load waited behind earlier stores would serialise the store stream.
template <int BN, int RA>
__device__ __forceinline__ void stage_epilogue_consts(float* ec, int tid, int nthreads, int64_t m0, int64_t n0,
                                                      int64_t M, int64_t N, const float* __restrict__ scale,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ rowadd, int64_t rows_per_batch,
                                                      float alpha, const float* __restrict__ oqp) {
    if (tid < 3) ec[(2 + RA) * BN + tid] = oqp ? oqp[tid] : 0.f;
    for (int c = tid; c < BN; c += nthreads) {
        const int64_t col = n0 + c;
        const bool in = col < N;
        ec[c] = in ? (scale ? scale[col] : alpha) : 0.f;
        ec[BN + c] = (in && bias) ? bias[col] : 0.f;
        const int64_t b0 = m0 / rows_per_batch;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int64_t b = b0 + j;
            ec[(2 + j) * BN + c] = (in && rowadd && b * rows_per_batch < M) ? rowadd[b * N + col] : 0.f;
        }
    }
}

template <int DT, int TM, int TN, int BN, int RA, int HG>
__device__ __forceinline__ void gemm_epilogue(typename Acc<DT>::type (&acc)[TM][TN], uint8_t* smem, const float* ec,
                                              int wave, int lane, int64_t m0, int64_t row0, int64_t col0, int ecol0,
                                              int64_t M, int64_t N, int64_t rows_per_batch, bool has_rowadd,
                                              const float* __restrict__ residual, int64_t ldr,
                                              float* __restrict__ out, int64_t ldo, int out_mode) {
    constexpr int EST = TN * 32 + 4;
    const int fr = lane & 31, fh = lane >> 5;
    const bool vec = ((N & 3) == 0) && ((ldo & 3) == 0) && (!residual || (ldr & 3) == 0) &&
                     ((((uintptr_t)out) & 15) == 0) && (!residual || (((uintptr_t)residual) & 15) == 0);
    __syncthreads();                                       // stage buffers free, epilogue constants visible
    if (vec) {
        float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EST);
        constexpr int C4 = TN * 8;
        constexpr int NIT = C4 / 2;                        // 32*C4 float4 per slab / 64 lanes
        const int64_t b0 = m0 / rows_per_batch;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[((r & 3) + 8 * (r >> 2) + 4 * fh) * EST + j * 32 + fr] = (float)acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int GRP = NIT / HG;                  // residual loads in flight per lane (HG rounds per slab)
#pragma unroll
            for (int h = 0; h < HG; ++h) {
                float4 rr[GRP];
#pragma unroll
                for (int u = 0; u < GRP; ++u) {            // every residual load of the round before its first store
                    const int idx = lane + 64 * (h * GRP + u);
                    const int rl = idx / C4, c4 = idx - rl * C4;
                    const int64_t row = row0 + i * 32 + rl, col = col0 + c4 * 4;
                    rr[u] = make_float4(0, 0, 0, 0);
                    if (residual && row < M && col < N)
                        rr[u] = *reinterpret_cast<const float4*>(residual + row * ldr + col);
                }
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const int idx = lane + 64 * (h * GRP + u);
                    const int rl = idx / C4, c4 = idx - rl * C4;
                    const int64_t row = row0 + i * 32 + rl, col = col0 + c4 * 4;
                    if (row >= M || col >= N) continue;
                    const int ecol = ecol0 + c4 * 4;
                    float4 v = *reinterpret_cast<const float4*>(ep + rl * EST + c4 * 4);
                    const float4 sc4 = *reinterpret_cast<const float4*>(ec + ecol);
                    const float4 b4 = *reinterpret_cast<const float4*>(ec + BN + ecol);
                    v.x = v.x * sc4.x + b4.x; v.y = v.y * sc4.y + b4.y; v.z = v.z * sc4.z + b4.z; v.w = v.w * sc4.w + b4.w;
                    if (has_rowadd) {
                        const int bj = (int)(row / rows_per_batch - b0);
                        const float4 a4 = *reinterpret_cast<const float4*>(ec + (2 + bj) * BN + ecol);
                        v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
                    }
                    v.x += rr[u].x; v.y += rr[u].y; v.z += rr[u].z; v.w += rr[u].w;
                    if (out_mode == 0) {
                        *reinterpret_cast<float4*>(out + row * ldo + col) = v;
                    } else {
                        // quantised outputs: the only consumer is an activation quantizer with fixed (delta, zp)
                        const float od = ec[(2 + RA) * BN], oz = ec[(2 + RA) * BN + 1], oq = ec[(2 + RA) * BN + 2];
                        const float oi = 1.0f / od;
                        if (out_mode == 3) {               // GEGLU on interleaved (a, gate) columns -> int8 operand
                            const float y0 = v.x * (0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752440f)));
                            const float y1 = v.z * (0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752440f)));
                            const int c0 = (int)fminf(fmaxf(rint_div(y0, od, oi) + oz, 0.f), oq) - 128;
                            const int c1 = (int)fminf(fmaxf(rint_div(y1, od, oi) + oz, 0.f), oq) - 128;
                            *reinterpret_cast<uint16_t*>(reinterpret_cast<int8_t*>(out) + row * ldo + (col >> 1)) =
                                (uint16_t)((c0 & 0xff) | ((c1 & 0xff) << 8));
                        } else {
                            const float q0 = fminf(fmaxf(rint_div(v.x, od, oi) + oz, 0.f), oq);
                            const float q1 = fminf(fmaxf(rint_div(v.y, od, oi) + oz, 0.f), oq);
                            const float q2 = fminf(fmaxf(rint_div(v.z, od, oi) + oz, 0.f), oq);
                            const float q3 = fminf(fmaxf(rint_div(v.w, od, oi) + oz, 0.f), oq);
                            if (out_mode == 1) {           // f16 operand code - zp (attention products)
                                __half2 h0 = __floats2half2_rn(q0 - oz, q1 - oz), h1 = __floats2half2_rn(q2 - oz, q3 - oz);
                                uint2 pk;
                                pk.x = *reinterpret_cast<uint32_t*>(&h0);
                                pk.y = *reinterpret_cast<uint32_t*>(&h1);
                                *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(out) + row * ldo + col) = pk;
                            } else {                       // int8 operand code - 128
                                const int a = (int)q0 - 128, b = (int)q1 - 128, c = (int)q2 - 128, d = (int)q3 - 128;
                                *reinterpret_cast<uint32_t*>(reinterpret_cast<int8_t*>(out) + row * ldo + col) =
                                    (uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8) | ((uint32_t)(c & 0xff) << 16) |
                                    ((uint32_t)(d & 0xff) << 24);
                            }
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
    const int64_t b0 = m0 / rows_per_batch;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t col = col0 + j * 32 + fr;
        if (col >= N) continue;
        const int ecol = ecol0 + j * 32 + fr;
        const float s = ec[ecol], bs = ec[BN + ecol];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row >= M) continue;
                float v = (float)acc[i][j][r] * s + bs;
                if (has_rowadd) v += ec[(2 + (int)(row / rows_per_batch - b0)) * BN + ecol];
                if (residual) v += residual[row * ldr + col];
                out[row * ldo + col] = v;
            }
        }
    }
}

// Source rows for everything that is not real data (convolution padding, M/N/K tails): row v holds
// 64 bytes of value v, so a direct-to-LDS load can fetch "padding" like any other address.
__device__ uint8_t g_pad_rows[256 * 64];
__global__ void k_init_pad_rows() {
    for (int i = threadIdx.x; i < 256 * 64; i += blockDim.x) g_pad_rows[i] = (uint8_t)(i >> 6);
}

template <int DT, int TM, int TN>
__global__ void __launch_bounds__(256)
k_gemm_nt(const uint8_t* __restrict__ A, int64_t lda_b, int64_t strideA_b, const uint8_t* __restrict__ Bm,
          int64_t ldb_b, int64_t strideB_b, int64_t M, int64_t N, int64_t Kb, ConvGeom g,
          const float* __restrict__ scale, const float* __restrict__ bias, const float* __restrict__ rowadd,
          int64_t rows_per_batch, const float* __restrict__ residual, int64_t ldr, float* __restrict__ out,
          int64_t ldo, int64_t strideC, float alpha, int inner, int64_t strideA_i, int64_t strideB_i,
          int64_t strideC_i, int out_mode, const float* __restrict__ oqp) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int NA = BM / 64, NB = BN / 64;   // 16-byte direct-to-LDS loads per thread per K-step
    constexpr int LPT = NA + NB;
    constexpr int KSTEP = 64;
    constexpr int STAGES = EDADM_GEMM_STAGES;
    constexpr int TILE = (BM + BN) * 64;
    constexpr int EST = TN * 32 + 4;                       // epilogue staging row stride (floats)
    constexpr int EPI_BYTES = 4 * 32 * EST * 4;
    constexpr int RA = BM / 16 + 1;                        // batch entries a tile can span (rows_per_batch >= 16)
    constexpr int MAIN_BYTES = STAGES * TILE > EPI_BYTES ? STAGES * TILE : EPI_BYTES;
    constexpr int SMEM_BYTES = MAIN_BYTES + (2 + RA) * BN * 4 + 16;
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM_BYTES];
    float* ec = reinterpret_cast<float*>(smem + MAIN_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    {   // batch index z = outer * inner + head: (batch, head) views of [B][N][heads*d] tensors
        const int64_t zo = blockIdx.z / inner, zi = blockIdx.z % inner;
        A += zo * strideA_b + zi * strideA_i;
        Bm += zo * strideB_b + zi * strideB_i;
        const int64_t coff = zo * strideC + zi * strideC_i;      // in output elements
        out = out_mode == 0 ? out + coff
              : out_mode == 1 ? reinterpret_cast<float*>(reinterpret_cast<__half*>(out) + coff)
                              : reinterpret_cast<float*>(reinterpret_cast<int8_t*>(out) + coff);
    }

    // ---- staging coordinates.  LDS image is lane-linear (thread t writes bytes [16t, 16t+16) of each
    // 4 KiB slab = row (t>>2)+64i, physical chunk t&3); the XOR swizzle is applied to the SOURCE chunk.
    const int sr = tid >> 2;
    const int sc = (tid & 3) ^ ((tid >> 4) & 3);      // logical chunk this thread fetches
    const uint8_t* zero_row = g_pad_rows;             // value 0
    const uint8_t* pad_row = g_pad_rows + (int)(uint8_t)g.padval * 64;
    int64_t a_base[NA];
    int a_y[NA], a_x[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int64_t m = m0 + sr + 64 * i;
        a_ok[i] = m < M;
        if (g.mode == 0) {
            a_base[i] = m * lda_b;
            a_y[i] = a_x[i] = 0;
        } else {
            const int64_t hw = (int64_t)g.Ho * g.Wo;
            const int64_t b = m / hw, r = m - b * hw;
            a_y[i] = (int)(r / g.Wo);
            a_x[i] = (int)(r - (int64_t)a_y[i] * g.Wo);
            a_base[i] = b * (int64_t)g.H * g.W;
        }
    }
    const uint8_t* b_row[NB];
    bool b_ok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int64_t n = n0 + sr + 64 * i;
        b_ok[i] = n < N;
        b_row[i] = Bm + (b_ok[i] ? n : 0) * ldb_b;
    }
    int tap_c = 0, ci_c = 0;
    if (g.mode != 0) { tap_c = (sc * 16) / g.Cin; ci_c = (sc * 16) % g.Cin; }
    const bool uniform_tap = g.mode != 0 && !g.ups && (g.Cin % KSTEP) == 0;
    int tap_s = 0, ci_s = 0;                        // wave-uniform (tap, channel) of the K-step, fast path
    int a_y0[NA], a_x0[NA], a_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        a_y0[i] = a_y[i] * g.stride - g.pad0;
        a_x0[i] = a_x[i] * g.stride - g.pad0;
        a_off[i] = (int)((a_base[i] + (int64_t)a_y0[i] * g.W + a_x0[i]) * g.Cin) + sc * 16;
    }

    auto issue_tile = [&](int stage, int64_t kb) {  // kb = byte offset along K
        const int64_t off = kb + sc * 16;
        const bool kin = off < Kb;
        const uint8_t* src[NA];
        if (g.mode == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) src[i] = (a_ok[i] && kin) ? A + a_base[i] + off : zero_row;
        } else if (uniform_tap) {
            // Cin % 64 == 0: the whole K-step lies in one tap -> tap arithmetic is scalar, a load costs
            // one 32-bit add, two compares and the pointer select
            const int ky = tap_s / g.KW, kx = tap_s - ky * g.KW;
            const int dlt = (ky * g.W + kx) * g.Cin + ci_s;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const bool in = (unsigned)(a_y0[i] + ky) < (unsigned)g.H && (unsigned)(a_x0[i] + kx) < (unsigned)g.W;
                src[i] = !(a_ok[i] && kin) ? zero_row : in ? A + (int64_t)(a_off[i] + dlt) : pad_row;
            }
            ci_s += 64;
            if (ci_s >= g.Cin) { ci_s = 0; ++tap_s; }
        } else {
            // this thread's chunk sits at k = kb + 16*sc: tap/ci tracked incrementally (Cin % 16 == 0 keeps a
            // chunk inside one tap)
            const int ky = tap_c / g.KW, kx = tap_c - ky * g.KW;
            const int Hl = g.ups ? 2 * g.H : g.H, Wl = g.ups ? 2 * g.W : g.W;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int iy = a_y[i] * g.stride + ky - g.pad0, ix = a_x[i] * g.stride + kx - g.pad0;
                const bool in = iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
                if (g.ups) { iy >>= 1; ix >>= 1; }
                src[i] = !(a_ok[i] && kin) ? zero_row
                         : in ? A + ((a_base[i] + (int64_t)iy * g.W + ix) * g.Cin + ci_c)
                              : pad_row;
            }
            ci_c += 64;
            while (ci_c >= g.Cin) { ci_c -= g.Cin; ++tap_c; }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
            glds16(src[i], lds0 + (uint32_t)(stage * TILE + i * 4096 + wave * 1024));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint8_t* s = (b_ok[i] && kin) ? b_row[i] + off : zero_row;
            glds16(s, lds0 + (uint32_t)(stage * TILE + BM * 64 + i * 4096 + wave * 1024));
        }
    };

    stage_epilogue_consts<BN, RA>(ec, tid, (int)blockDim.x, m0, n0, M, N, scale, bias, rowadd, rows_per_batch, alpha, oqp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // keep the pipeline's vmcnt bookkeeping to LDS-DMA only

    typename Acc<DT>::type acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int64_t nk = (Kb + 63) / 64;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p)
        if (p < nk) issue_tile(p, (int64_t)p * 64);
    for (int64_t kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the newer tile's LPT loads are still in flight
        // tiles kt+1 .. kt+STAGES-2 may still be in flight
        const int64_t ahead = nk - 1 - kt < STAGES - 2 ? nk - 1 - kt : STAGES - 2;
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + STAGES - 1 < nk) issue_tile((int)((kt + STAGES - 1) % STAGES), (kt + STAGES - 1) * 64);
        const uint8_t* As = smem + (int)(kt % STAGES) * TILE;
        const uint8_t* Bs = As + BM * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = 2 * ks + fh;
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (TM * 32) + i * 32 + fr;
                fa[i] = *reinterpret_cast<const uint4*>(As + (r * 4 + (c ^ ((r >> 2) & 3))) * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (TN * 32) + j * 32 + fr;
                fb[j] = *reinterpret_cast<const uint4*>(Bs + (r * 4 + (c ^ ((r >> 2) & 3))) * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    mma_step<DT>(fa[i], fb[j], acc[i][j]);
                }
        }
    }

    gemm_epilogue<DT, TM, TN, BN, RA, (TN % 2 == 0 || TN == 3) ? 4 : 2>(acc, smem, ec, wave, lane, m0, m0 + wm * (TM * 32), n0 + wn * (TN * 32),
                                      wn * (TN * 32), M, N, rows_per_batch, rowadd != nullptr, residual, ldr, out, ldo, out_mode);
}

// ---- 8-wave variant for the large-M layers: 256 x (64*TN) tile, 128-byte K rows (full cache lines per
// request, half the L2->LDS bytes per flop of the 128-row tile), two LDS stages, 24 MFMAs per wave per
// barrier.  Same gather / padding / epilogue contract as k_gemm_nt.
template <int DT, int TN>
__global__ void __launch_bounds__(512)
k_gemm_nt8(const uint8_t* __restrict__ A, int64_t lda_b, int64_t strideA_b, const uint8_t* __restrict__ Bm,
           int64_t ldb_b, int64_t strideB_b, int64_t M, int64_t N, int64_t Kb, ConvGeom g,
           const float* __restrict__ scale, const float* __restrict__ bias, const float* __restrict__ rowadd,
           int64_t rows_per_batch, const float* __restrict__ residual, int64_t ldr, float* __restrict__ out,
           int64_t ldo, int64_t strideC, float alpha, int inner, int64_t strideA_i, int64_t strideB_i,
           int64_t strideC_i, int out_mode, const float* __restrict__ oqp) {
    constexpr int TM = 2;
    constexpr int BM = 256, BN = 64 * TN;
    constexpr int NA = 4, NB = TN;                 // 64-row passes per operand (512 threads x 16 B = 64 rows x 128 B)
    constexpr int STAGES = 2;
    constexpr int KSTEP = 128;
    constexpr int TILE = (BM + BN) * 128;
    constexpr int EST = TN * 32 + 4;
    constexpr int EPI_BYTES = 8 * 32 * EST * 4;
    constexpr int RA = BM / 16 + 1;
    constexpr int MAIN_BYTES = STAGES * TILE > EPI_BYTES ? STAGES * TILE : EPI_BYTES;
    constexpr int SMEM_BYTES = MAIN_BYTES + (2 + RA) * BN * 4 + 16;
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM_BYTES];
    float* ec = reinterpret_cast<float*>(smem + MAIN_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    {
        const int64_t zo = blockIdx.z / inner, zi = blockIdx.z % inner;
        A += zo * strideA_b + zi * strideA_i;
        Bm += zo * strideB_b + zi * strideB_i;
        const int64_t coff = zo * strideC + zi * strideC_i;      // in output elements
        out = out_mode == 0 ? out + coff
              : out_mode == 1 ? reinterpret_cast<float*>(reinterpret_cast<__half*>(out) + coff)
                              : reinterpret_cast<float*>(reinterpret_cast<int8_t*>(out) + coff);
    }
    const int sr = tid >> 3;
    const int sc = (tid & 7) ^ ((tid >> 4) & 7);      // source chunk = physical chunk ^ ((row >> 1) & 7)
    const uint8_t* zero_row = g_pad_rows;
    const uint8_t* pad_row = g_pad_rows + (int)(uint8_t)g.padval * 64;
    int64_t a_base[NA];
    int a_y[NA], a_x[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int64_t m = m0 + sr + 64 * i;
        a_ok[i] = m < M;
        if (g.mode == 0) {
            a_base[i] = m * lda_b;
            a_y[i] = a_x[i] = 0;
        } else {
            const int64_t hw = (int64_t)g.Ho * g.Wo;
            const int64_t b = m / hw, r = m - b * hw;
            a_y[i] = (int)(r / g.Wo);
            a_x[i] = (int)(r - (int64_t)a_y[i] * g.Wo);
            a_base[i] = b * (int64_t)g.H * g.W;
        }
    }
    const uint8_t* b_row[NB];
    bool b_ok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int64_t n = n0 + sr + 64 * i;
        b_ok[i] = n < N;
        b_row[i] = Bm + (b_ok[i] ? n : 0) * ldb_b;
    }
    int tap_c = 0, ci_c = 0;
    if (g.mode != 0) { tap_c = (sc * 16) / g.Cin; ci_c = (sc * 16) % g.Cin; }
    const bool uniform_tap = g.mode != 0 && !g.ups && (g.Cin % KSTEP) == 0;
    int tap_s = 0, ci_s = 0;                        // wave-uniform (tap, channel) of the K-step, fast path
    int a_y0[NA], a_x0[NA], a_off[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        a_y0[i] = a_y[i] * g.stride - g.pad0;
        a_x0[i] = a_x[i] * g.stride - g.pad0;
        a_off[i] = (int)((a_base[i] + (int64_t)a_y0[i] * g.W + a_x0[i]) * g.Cin) + sc * 16;
    }

    auto issue_tile = [&](int stage, int64_t kb) {
        const int64_t off = kb + sc * 16;
        const bool kin = off < Kb;
        const uint8_t* src[NA];
        if (g.mode == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) src[i] = (a_ok[i] && kin) ? A + a_base[i] + off : zero_row;
        } else if (uniform_tap) {
            // Cin % 128 == 0: the whole K-step lies in one tap -> tap arithmetic is scalar, a load costs
            // one 32-bit add, two compares and the pointer select
            const int ky = tap_s / g.KW, kx = tap_s - ky * g.KW;
            const int dlt = (ky * g.W + kx) * g.Cin + ci_s;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const bool in = (unsigned)(a_y0[i] + ky) < (unsigned)g.H && (unsigned)(a_x0[i] + kx) < (unsigned)g.W;
                src[i] = !(a_ok[i] && kin) ? zero_row : in ? A + (int64_t)(a_off[i] + dlt) : pad_row;
            }
            ci_s += 128;
            if (ci_s >= g.Cin) { ci_s = 0; ++tap_s; }
        } else {
            const int ky = tap_c / g.KW, kx = tap_c - ky * g.KW;
            const int Hl = g.ups ? 2 * g.H : g.H, Wl = g.ups ? 2 * g.W : g.W;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int iy = a_y[i] * g.stride + ky - g.pad0, ix = a_x[i] * g.stride + kx - g.pad0;
                const bool in = iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
                if (g.ups) { iy >>= 1; ix >>= 1; }
                src[i] = !(a_ok[i] && kin) ? zero_row
                         : in ? A + ((a_base[i] + (int64_t)iy * g.W + ix) * g.Cin + ci_c)
                              : pad_row;
            }
            ci_c += 128;
            while (ci_c >= g.Cin) { ci_c -= g.Cin; ++tap_c; }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) glds16(src[i], lds0 + (uint32_t)(stage * TILE + i * 8192 + wave * 1024));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint8_t* s = (b_ok[i] && kin) ? b_row[i] + off : zero_row;
            glds16(s, lds0 + (uint32_t)(stage * TILE + BM * 128 + i * 8192 + wave * 1024));
        }
    };

    stage_epilogue_consts<BN, RA>(ec, tid, (int)blockDim.x, m0, n0, M, N, scale, bias, rowadd, rows_per_batch, alpha, oqp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // keep the pipeline's vmcnt bookkeeping to LDS-DMA only

    typename Acc<DT>::type acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int64_t nk = (Kb + 127) / 128;
    const int fr = lane & 31, fh = lane >> 5;
    issue_tile(0, 0);
    for (int64_t kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) issue_tile((int)((kt + 1) & 1), (kt + 1) * 128);
        const uint8_t* As = smem + (int)(kt & 1) * TILE;
        const uint8_t* Bs = As + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = 2 * ks + fh;
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * 64 + i * 32 + fr;
                fa[i] = *reinterpret_cast<const uint4*>(As + r * 128 + ((c ^ ((r >> 1) & 7)) * 16));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (TN * 32) + j * 32 + fr;
                fb[j] = *reinterpret_cast<const uint4*>(Bs + r * 128 + ((c ^ ((r >> 1) & 7)) * 16));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    mma_step<DT>(fa[i], fb[j], acc[i][j]);
                }
        }
    }
    gemm_epilogue<DT, TM, TN, BN, RA, 1>(acc, smem, ec, wave, lane, m0, m0 + wm * 64, n0 + wn * (TN * 32), wn * (TN * 32),
                                      M, N, rows_per_batch, rowadd != nullptr, residual, ldr, out, ldo, out_mode);
}

template <int DT>
static int launch_gemm(const void* A, int64_t lda_b, int64_t sA, const void* Bm, int64_t ldb_b, int64_t sB,
                       int64_t M, int64_t N, int64_t Kb, const ConvGeom& g, const float* scale, const float* bias,
                       const float* rowadd, int64_t rpb, const float* residual, int64_t ldr, float* out, int64_t ldo,
                       int64_t sC, int64_t batch, float alpha, hipStream_t st, int inner = 1, int64_t sAi = 0,
                       int64_t sBi = 0, int64_t sCi = 0, int out_mode = 0, const float* oqp = nullptr) {
    if (out_mode != 0 && (!oqp || (N & 3) || (ldo & 3))) return EDADM_EINVAL;   // quantised outputs use the 16-byte path
    if (!rowadd) rpb = M;                                    // one (unused) batch entry
    static bool pad_ready = false;
    if (!pad_ready) {       // stream-ordered ahead of the first GEMM; idempotent if it lands inside a captured graph
        hipLaunchKernelGGL(k_init_pad_rows, dim3(1), dim3(256), 0, st);
        pad_ready = true;
    }
    // tile choice: widest N tile that divides N well (192 for the 192-multiples of LDM-4, else 128, 64)
    int tn = 2;
    if (N % 192 == 0) tn = 3;
    else if (N <= 64) tn = 1;
    int tm = 2;
    if (M <= 64) tm = 1;
    // large-M layers: 256-row, 8-wave tile when it still fills the 256 CUs
    const int64_t tiles8 = ((M + 255) / 256) * ((N + 64 * tn - 1) / (64 * tn)) * batch;
    // convolutions whose Cin is a multiple of 64 but not of 128 keep scalar tap arithmetic only with 64-byte K-steps
    const bool nt8_gather_ok = true;
    static const int force = getenv("EDADM_GEMM_FORCE") ? atoi(getenv("EDADM_GEMM_FORCE")) : 0;   // diagnostics only
    if (force != 2 && EDADM_USE_NT8 && (force == 3 || (tiles8 >= 224 && Kb >= 256)) && nt8_gather_ok) {
        const dim3 grid8((unsigned)((N + 64 * tn - 1) / (64 * tn)), (unsigned)((M + 255) / 256), (unsigned)batch);
#define EDADM_GEMM8_CASE(TN_)                                                                                  \
        if (tn == TN_) {                                                                                       \
            hipLaunchKernelGGL((k_gemm_nt8<DT, TN_>), grid8, dim3(512), 0, st, (const uint8_t*)A, lda_b, sA,   \
                               (const uint8_t*)Bm, ldb_b, sB, M, N, Kb, g, scale, bias, rowadd, rpb, residual, \
                               ldr, out, ldo, sC, alpha, inner, sAi, sBi, sCi, out_mode, oqp);                 \
            return edadm_launch_status();                                                                      \
        }
        EDADM_GEMM8_CASE(3)
        EDADM_GEMM8_CASE(2)
        EDADM_GEMM8_CASE(1)
#undef EDADM_GEMM8_CASE
    }
    const dim3 blk(256);
#define EDADM_GEMM_CASE(TM_, TN_)                                                                              \
    if (tm == TM_ && tn == TN_) {                                                                              \
        const dim3 grid((unsigned)((N + 64 * TN_ - 1) / (64 * TN_)), (unsigned)((M + 64 * TM_ - 1) / (64 * TM_)), \
                        (unsigned)batch);                                                                      \
        hipLaunchKernelGGL((k_gemm_nt<DT, TM_, TN_>), grid, blk, 0, st, (const uint8_t*)A, lda_b, sA,          \
                           (const uint8_t*)Bm, ldb_b, sB, M, N, Kb, g, scale, bias, rowadd, rpb, residual, ldr, \
                           out, ldo, sC, alpha, inner, sAi, sBi, sCi, out_mode, oqp);                          \
        return edadm_launch_status();                                                                          \
    }
    EDADM_GEMM_CASE(2, 3)
    EDADM_GEMM_CASE(2, 2)
    EDADM_GEMM_CASE(2, 1)
    EDADM_GEMM_CASE(1, 3)
    EDADM_GEMM_CASE(1, 2)
    EDADM_GEMM_CASE(1, 1)
#undef EDADM_GEMM_CASE
    return EDADM_EINVAL;
}

extern "C" int edadm_qgemm_i8(const int8_t* A, int64_t lda, const int8_t* Wt, int64_t ldw, int64_t M, int64_t N,
                              int64_t K, const int32_t* geom, const float* scale, const float* bias,
                              const float* rowadd, int64_t rows_per_batch, const float* residual, int64_t ldr,
                              float* out, int64_t ldo, void* stream) {
    if (!A || !Wt || !out || !scale || M <= 0 || N <= 0 || K <= 0 || (K & 15) || (ldw & 15)) return EDADM_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)Wt & 15)) return EDADM_EINVAL;
    ConvGeom g;
    if (geom) {
        const int32_t* p = geom;
        g = ConvGeom{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], 0, 0, 0};
    } else {
        g = ConvGeom{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    }
    if (g.mode == 0) {
        if (lda & 15) return EDADM_EINVAL;
    } else {
        if (g.mode != 1 || (g.Cin & 15) || (int64_t)g.KH * g.KW * g.Cin != K || g.stride < 1 ||
            (int64_t)g.B * g.Ho * g.Wo != M)
            return EDADM_EINVAL;
    }
    if (rowadd && rows_per_batch < 16) return EDADM_EINVAL;   // the epilogue stages <= BM/16+1 row-add rows in LDS
    return launch_gemm<0>(A, lda, 0, Wt, ldw, 0, M, N, K, g, scale, bias, rowadd, rows_per_batch, residual, ldr,
                             out, ldo, 0, 1, 1.0f, (hipStream_t)stream);
}

extern "C" int edadm_gemm_f16_nt(const void* A, int64_t lda, int64_t strideA, int64_t strideA_i, const void* Bm,
                                 int64_t ldb, int64_t strideB, int64_t strideB_i, float* C, int64_t ldc,
                                 int64_t strideC, int64_t strideC_i, int64_t batch, int64_t inner, int64_t M,
                                 int64_t N, int64_t K, float alpha, void* stream) {
    if (!A || !Bm || !C || batch <= 0 || inner <= 0 || M <= 0 || N <= 0 || K <= 0 || (K & 7) || (lda & 7) ||
        (ldb & 7) || (strideA & 7) || (strideB & 7) || (strideA_i & 7) || (strideB_i & 7))
        return EDADM_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)Bm & 15)) return EDADM_EINVAL;
    ConvGeom g{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return launch_gemm<1>(A, lda * 2, strideA * 2, Bm, ldb * 2, strideB * 2, M, N, K * 2, g, nullptr, nullptr,
                              nullptr, 1, nullptr, 0, C, ldc, strideC, batch * inner, alpha, (hipStream_t)stream,
                              (int)inner, strideA_i * 2, strideB_i * 2, strideC_i);
}

// same contract as edadm_qgemm_i8 with f16 operands (a = code - zp_x, w = wcode - zp_w as exact f16 integers):
// used for layers whose integer weight range does not fit int8 (8-bit weights with zp 127).
extern "C" int edadm_qgemm_f16(const void* A, int64_t lda, const void* Wt, int64_t ldw, int64_t M, int64_t N,
                               int64_t K, const int32_t* geom, const float* scale, const float* bias,
                               const float* rowadd, int64_t rows_per_batch, const float* residual, int64_t ldr,
                               float* out, int64_t ldo, void* stream) {
    if (!A || !Wt || !out || !scale || M <= 0 || N <= 0 || K <= 0 || (K & 7) || (ldw & 7)) return EDADM_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)Wt & 15)) return EDADM_EINVAL;
    ConvGeom g{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (geom) {
        const int32_t* p = geom;
        // byte-addressed gather: one pixel is Cin f16 = 2*Cin bytes
        g = ConvGeom{p[0], p[1], p[2], p[3], p[4] * 2, p[5], p[6], p[7], p[8], p[9], p[10], p[11], 0, 0, 0, 0};
        if (g.mode != 1 || (g.Cin & 15) || (int64_t)g.KH * g.KW * p[4] != K || (int64_t)g.B * g.Ho * g.Wo != M)
            return EDADM_EINVAL;
    } else if (lda & 7) {
        return EDADM_EINVAL;
    }
    if (rowadd && rows_per_batch < 16) return EDADM_EINVAL;   // the epilogue stages <= BM/16+1 row-add rows in LDS
    return launch_gemm<1>(A, lda * 2, 0, Wt, ldw * 2, 0, M, N, K * 2, g, scale, bias, rowadd, rows_per_batch,
                              residual, ldr, out, ldo, 0, 1, 1.0f, (hipStream_t)stream);
}

// ---- variants whose epilogue feeds an activation quantizer directly (no fp32 round trip through HBM):
// out_mode 1: f16 operand (code - zp), 2: int8 operand (code - 128), 3: GEGLU over interleaved (a, gate)
// output columns then int8 operand [M][N/2] (attention.py:37-45 + the consumer's quantizer,
// quant_layer.py:266-269).  oqp = device float[3] {delta, zp, qmax} of the consuming quantizer.
extern "C" int edadm_qgemm_i8_q(const int8_t* A, int64_t lda, const int8_t* Wt, int64_t ldw, int64_t M, int64_t N,
                                int64_t K, const int32_t* geom, const float* scale, const float* bias,
                                const float* rowadd, int64_t rows_per_batch, const float* residual, int64_t ldr,
                                void* out, int64_t ldo, int out_mode, const float* oqp, void* stream) {
    if (!A || !Wt || !out || !scale || M <= 0 || N <= 0 || K <= 0 || (K & 15) || (ldw & 15)) return EDADM_EINVAL;
    if (out_mode < 1 || out_mode > 3 || !oqp) return EDADM_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)Wt & 15)) return EDADM_EINVAL;
    ConvGeom g{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (geom) {
        const int32_t* p = geom;
        g = ConvGeom{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], 0, 0, 0};
        if (g.mode != 1 || (g.Cin & 15) || (int64_t)g.KH * g.KW * g.Cin != K || g.stride < 1 ||
            (int64_t)g.B * g.Ho * g.Wo != M)
            return EDADM_EINVAL;
    } else if (lda & 15) {
        return EDADM_EINVAL;
    }
    if (rowadd && rows_per_batch < 16) return EDADM_EINVAL;
    return launch_gemm<0>(A, lda, 0, Wt, ldw, 0, M, N, K, g, scale, bias, rowadd, rows_per_batch, residual, ldr,
                             (float*)out, ldo, 0, 1, 1.0f, (hipStream_t)stream, 1, 0, 0, 0, out_mode, oqp);
}

extern "C" int edadm_gemm_f16_nt_q(const void* A, int64_t lda, int64_t strideA, int64_t strideA_i, const void* Bm,
                                   int64_t ldb, int64_t strideB, int64_t strideB_i, void* C, int64_t ldc,
                                   int64_t strideC, int64_t strideC_i, int64_t batch, int64_t inner, int64_t M,
                                   int64_t N, int64_t K, float alpha, int out_mode, const float* oqp, void* stream) {
    if (!A || !Bm || !C || batch <= 0 || inner <= 0 || M <= 0 || N <= 0 || K <= 0 || (K & 7) || (lda & 7) ||
        (ldb & 7) || (strideA & 7) || (strideB & 7) || (strideA_i & 7) || (strideB_i & 7))
        return EDADM_EINVAL;
    if (out_mode < 1 || out_mode > 2 || !oqp) return EDADM_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)Bm & 15)) return EDADM_EINVAL;
    ConvGeom g{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return launch_gemm<1>(A, lda * 2, strideA * 2, Bm, ldb * 2, strideB * 2, M, N, K * 2, g, nullptr, nullptr,
                              nullptr, 1, nullptr, 0, (float*)C, ldc, strideC, batch * inner, alpha, (hipStream_t)stream,
                              (int)inner, strideA_i * 2, strideB_i * 2, strideC_i, out_mode, oqp);
}

// ---- fp32 NT GEMM on v_mfma_f32_32x32x2_f32 for the calibration graph (H1): C[z] = alpha * A[z] . B[z]^T
// (+ bias[n]) (+ residual[m][n]).  Exact fp32 FMA chains (no reduced-precision path exists on gfx950).
// quant_layer.py:434 (F.conv2d / F.linear on fake-quantised operands) and its autograd backward are built
// from this entry point plus the im2col / col2im / transpose / slab-sum kernels below.
extern "C" int edadm_gemm_f32_nt(const float* A, int64_t lda, int64_t strideA, const float* Bm, int64_t ldb,
                                 int64_t strideB, float* C, int64_t ldc, int64_t strideC, int64_t batch, int64_t M,
                                 int64_t N, int64_t K, float alpha, const float* bias, const float* residual,
                                 int64_t ldr, void* stream) {
    if (!A || !Bm || !C || batch <= 0 || M <= 0 || N <= 0 || K <= 0 || (K & 3) || (lda & 3) || (ldb & 3) ||
        (strideA & 3) || (strideB & 3))
        return EDADM_EINVAL;
    if (((uintptr_t)A & 15) || ((uintptr_t)Bm & 15)) return EDADM_EINVAL;
    if (residual && batch != 1) return EDADM_EINVAL;
    ConvGeom g{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return launch_gemm<2>(A, lda * 4, strideA * 4, Bm, ldb * 4, strideB * 4, M, N, K * 4, g, nullptr, bias, nullptr, 1,
                          residual, ldr, C, ldc, strideC, batch, alpha, (hipStream_t)stream);
}
