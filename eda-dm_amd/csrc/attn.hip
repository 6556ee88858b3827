// K6f -- fused quantised attention for the multi-head / long-sequence case (quant_block.py:204-235 QuantBasicTransformerBlock
// attention, :119-162 QuantQKMatMul / QuantSMVMatMul of the legacy AttentionBlock, :398-451 QuantAttnBlock):
//
//     S = alpha_qk * Qc Kc^T        Qc, Kc, Vc: integer codes minus zero point, exact in f16 (|v| <= 255)
//     P = softmax(S)                fp32, row maximum subtracted (torch.softmax)
//     Pc = clamp(rint(P / dw) + zw, 0, qmax) - zw
//     O = alpha_pv * Pc Vc          (alpha_pv = dw dv)
//
// in ONE kernel: the score matrix never exists in memory (the three-kernel path writes heads x Nq x Nk fp32 scores and reads
// them back: 4.3 GB per 64x64 self-attention of Stable Diffusion at 8 rows, the dominant cost of configs 3 and 5).  The
// probability codes need the FINAL row maximum and row sum before the first one can be rounded, so the kernel walks the keys
// twice -- (1) row maximum and sum of exp(s - max) in the online form, (2) codes and Pc Vc -- recomputing S on the f16 MFMA each time
// (the head dimension is small: the products are cheap; the kernel is VALU-bound on the exponentials and the rounding, which is
// why both are the lean forms: hardware exp2, reciprocal multiply with an exact fallback next to a rounding boundary).  All
// integer products are exact in fp32 (|Qc Kc| summed over d <= 160 stays below 2^24; codes of a probability row sum to ~255), so
// against the three-kernel path the result differs only through the fp32 row sum's order and the exponential's last bit: a
// probability code that sits on a rounding boundary may land on the other side (measured: 0.003-0.06 % of outputs, one code).
//
// Layout trick: the kernel computes S^T = Kc Qc^T, so that in the MFMA accumulator layout (32x32: lane -> column, 16 rows) a
// LANE OWNS ONE QUERY and 16 keys of every 32-key block: row maximum and row sum are per-lane scalars (one cross-lane
// exchange at the end of a pass), and the 16 probability codes of a lane, packed to f16, ARE the B operand of the P V product
// for two 16-key K-steps if the keys of a K-step are taken in the order the accumulator rows have (lane half h holds keys
// {4h .. 4h+3, 8+4h .. 8+4h+3} of each 16): V^T is fetched from LDS in that order, P never moves between registers.
#include "common.h"
// Softmax codes rint((e / sum) / delta) through ONE reciprocal: t = fl(e * fl(1 / fl(sum * delta))) is within 3 x 2^-24 |t| of the quotient,
// the reference's fl(fl(e / sum) / delta) within 2 x 2^-24: the rounded integers can differ only inside the band |t - rint(t)| +
// 3.6e-7 t > 0.5 - 4e-5, whose lanes redo both divisions (relative: tight for 8-bit codes, sufficient for 16-bit ones; common.h)
#define EDADM_SM_BAND_REL 3.6e-7f
#define EDADM_SM_NEAR (0.5f - 4e-5f)
#include "../../include/edadm.h"
#include <hip/hip_fp16.h>
#include <type_traits>
#include <stdlib.h>
// launch choices are compile-time constants in the product; diagnostic builds (-DEDADM_DIAG: tools/attn_wide_bench.hip) read them
// from the environment
#ifdef EDADM_DIAG
#define EDADM_TUNE_I(name, dflt) (getenv(name) ? atoll(getenv(name)) : (long long)(dflt))
#else
#define EDADM_TUNE_I(name, dflt) ((long long)(dflt))
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

// quantiser table entry as every producer of the ABI reads it (quant_norm.hip): {delta, zero point, qmax, -}
struct QP { float d, z, qmax, inv; };
__device__ __forceinline__ QP qp_load(const QP* p, int i) {
    QP q = p[i];
    q.inv = 1.0f / q.d;
    return q;
}

#define ATT_BQ 128            // queries per workgroup: 4 waves x 32
#define ATT_BK 64             // keys per staged block: 2 x 32

// KD: padded head dimension / 16 (K-steps of Q K^T); DVB: 32-wide blocks of the output head dimension
template <int KD, int DVB>
__global__ void __launch_bounds__(256)
k_attn_fused(const __half* __restrict__ Q, int64_t ldq, int64_t sQ, int64_t hQ, const __half* __restrict__ K, int64_t ldk,
             int64_t sK, int64_t hK, const __half* __restrict__ V, int64_t ldv, int64_t sV, int64_t hV, void* __restrict__ out,
             int64_t ldo, int64_t sO,
             int Nq, int Nk, int d, float alpha_qk, const QP* __restrict__ pqp, float alpha_pv, int out_mode,
             const QP* __restrict__ oqp) {
    const QP pw = qp_load(pqp, 0);
    constexpr int DP = KD * 16;                   // padded head dimension
    constexpr int KROW = DP + 8;                  // LDS row of a key (halfs): +16 bytes spreads the banks
    constexpr int VROW = ATT_BK + 8;              // LDS row of V^T (one output dimension, ATT_BK keys)
    constexpr int DVP = DVB * 32;
    __shared__ __half lk_[2][ATT_BK * KROW];               // two buffers: the next block is written while this one is read,
    __shared__ __half lv_[2][DVP * VROW];                  // one barrier per block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int q = blockIdx.x * ATT_BQ + wave * 32 + fr;          // this lane's query
    const __half* Qb = Q + b * sQ + (int64_t)h * hQ;             // head strides: d for [.., heads*d], 3d for the legacy (q|k|v) layout
    const __half* Kb = K + b * sK + (int64_t)h * hK;
    const __half* Vb = V + b * sV + (int64_t)h * hV;

    // Q fragments (B operand of S^T = K Q^T): lane -> query fr, 8 head dimensions 16 ks + 8 fh .. + 7
    half8 qf[KD];
#pragma unroll
    for (int ks = 0; ks < KD; ++ks) {
        const int c = ks * 16 + fh * 8;
        half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q < Nq && c < d) z = *reinterpret_cast<const half8*>(Qb + (int64_t)q * ldq + c);
        qf[ks] = z;
    }
    const int nkb = (Nk + ATT_BK - 1) / ATT_BK;
    constexpr int KCH = DP / 8;                                   // 16-byte chunks per key row
    constexpr int KPT = (ATT_BK * KCH + 255) / 256;               // chunks per thread and block
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 rk[KPT], rv[KPT];                                       // the NEXT block on its way from L2 while this one computes

    auto gload = [&](int kb, bool with_v) {
        const int k0 = kb * ATT_BK;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * 256;
            const int key = idx / KCH, ch = idx - key * KCH;
            const bool ok = idx < ATT_BK * KCH && (k0 + key) < Nk && ch * 8 < d;
            rk[i] = ok ? *reinterpret_cast<const half8*>(Kb + (int64_t)(k0 + key) * ldk + ch * 8) : zero8;
            if (with_v) rv[i] = ok ? *reinterpret_cast<const half8*>(Vb + (int64_t)(k0 + key) * ldv + ch * 8) : zero8;
        }
    };
    auto lstore = [&](int buf, bool with_v) {
        __half* lk = lk_[buf];
        __half* lv = lv_[buf];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * 256;
            if (idx < ATT_BK * KCH) {
                const int key = idx / KCH, ch = idx - key * KCH;
                *reinterpret_cast<half8*>(lk + key * KROW + ch * 8) = rk[i];
                if (with_v && ch * 8 < DVP) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) lv[(ch * 8 + e) * VROW + key] = (__half)rv[i][e];
                }
            }
        }
    };
    // S^T of one staged block for this wave: 2 sub-blocks of 32 keys x this lane's query; register r of sub-block sb is key
    // 32 sb + 8 (r / 4) + 4 fh + (r % 4)
    auto scores = [&](const __half* lk, float16v (&acc)[2]) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            float16v c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KD; ++ks) {
                const half8 a = *reinterpret_cast<const half8*>(lk + (sb * 32 + fr) * KROW + ks * 16 + fh * 8);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], c, 0, 0, 0);
            }
            acc[sb] = c;
        }
    };
    auto key_of = [&](int kb, int sb, int r) { return kb * ATT_BK + sb * 32 + 8 * (r >> 2) + 4 * fh + (r & 3); };
    // one walk over the keys: block kb is computed from LDS buffer kb & 1 while block kb + 1 (loaded from L2 during block kb - 1)
    // is written into the other buffer and block kb + 2 is requested: one barrier per block
    auto walk = [&](bool with_v, auto&& body) {
        gload(0, with_v);
        lstore(0, with_v);
        __syncthreads();
        if (nkb > 1) gload(1, with_v);
        for (int kb = 0; kb < nkb; ++kb) {
            float16v acc[2];
            scores(lk_[kb & 1], acc);
            body(kb, acc, (kb + 1) * ATT_BK <= Nk, lv_[kb & 1]);
            if (kb + 1 < nkb) lstore((kb + 1) & 1, with_v);
            __syncthreads();
            if (kb + 2 < nkb) gload(kb + 2, with_v);
        }
    };

    // ---- walk 1: row maximum and row sum together (online form): the running sum is rescaled when the running maximum of the
    // raw integer products (alpha_qk > 0) grows.  exp(alpha s - alpha max) is exp2(c s - c max), c = alpha log2(e), on the hardware
    // exponential (v_exp_f32, 1 ulp): the accuracy class of the device exp a GPU softmax runs on
    const float cexp = alpha_qk * 1.44269504088896340736f;
    float mx = -INFINITY, sum = 0.f;
    walk(false, [&](int kb, float16v (&acc)[2], bool full, const __half*) {
        float bm = mx;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!full && key_of(kb, sb, r) >= Nk) acc[sb][r] = -INFINITY;
                bm = fmaxf(bm, acc[sb][r]);
            }
        if (bm == -INFINITY) return;                               // nothing valid yet for this lane (only past the last key)
        sum *= __builtin_amdgcn_exp2f((mx - bm) * cexp);           // mx = -inf the first time: exp2(-inf) = 0, sum is 0 anyway
        mx = bm;
        const float cmax = mx * cexp;
        float part = 0.f;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) part += __builtin_amdgcn_exp2f(fmaf(acc[sb][r], cexp, -cmax));
        sum += part;
    });
    {   // the two lanes of a query (16 keys of every 32 each) combine
        const float mo = __shfl_xor(mx, 32, 64), so = __shfl_xor(sum, 32, 64);
        const float m = fmaxf(mx, mo);
        sum = sum * __builtin_amdgcn_exp2f((mx - m) * cexp) + so * __builtin_amdgcn_exp2f((mo - m) * cexp);
        mx = m;
    }
    const float cmax = mx * cexp;
    const float inv = 1.0f / (sum * pw.d);                         // e / sum / delta ~ e * inv; boundary cases redo both divisions
    const bool z0 = pw.z == 0.f;                                   // always_zero quantisers (every softmax quantiser of the reference)
    // ---- walk 2: probability codes and O^T = V^T P^T
    float16v o[DVB];
#pragma unroll
    for (int j = 0; j < DVB; ++j) o[j] = float16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    walk(true, [&](int kb, float16v (&acc)[2], bool full, const __half* lv) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            float r_[16];
            float worst = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = __builtin_amdgcn_exp2f(fmaf(acc[sb][r], cexp, -cmax));
                if (!full) e = key_of(kb, sb, r) < Nk ? e : 0.f;
                acc[sb][r] = e;
                const float t = e * inv;
                r_[r] = rintf(t);
                worst = fmaxf(worst, fmaf(t, EDADM_SM_BAND_REL, fabsf(t - r_[r])));
            }
            if (__builtin_expect(worst > EDADM_SM_NEAR, 0)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    asm volatile("" : "+v"(r_[r]));
                    r_[r] = rintf((acc[sb][r] / sum) / pw.d);
                }
            }
            half8 pf[2];
            if (z0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pf[r >> 3][r & 7] = (_Float16)fminf(r_[r], pw.qmax);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) pf[r >> 3][r & 7] = (_Float16)(fminf(fmaxf(r_[r] + pw.z, 0.f), pw.qmax) - pw.z);
            }
            // two K-steps of 16 keys: lane half fh supplies keys {4 fh .. +3, 8 + 4 fh .. +3} of each -- the accumulator's own order
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int kofs = sb * 32 + j2 * 16 + 4 * fh;
#pragma unroll
                for (int j = 0; j < DVB; ++j) {
                    const __half* vp = lv + (j * 32 + fr) * VROW + kofs;
                    const half4 v0 = *reinterpret_cast<const half4*>(vp), v1 = *reinterpret_cast<const half4*>(vp + 8);
                    const half8 a = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[j2], o[j], 0, 0, 0);
                }
            }
        }
    });
    if (q >= Nq) return;
    // ---- epilogue: lane -> query q, output dimensions 32 j + 8 g + 4 fh + e
    QP oq;
    if (out_mode == 2) oq = qp_load(oqp, 0);
#pragma unroll
    for (int j = 0; j < DVB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dv = j * 32 + 8 * g + 4 * fh;
            if (dv >= d) continue;
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = o[j][4 * g + e] * alpha_pv;
            const int64_t col = (int64_t)h * d + dv;
            if (out_mode == 0) {
                float* op = reinterpret_cast<float*>(out) + b * sO + (int64_t)q * ldo + col;
                *reinterpret_cast<float4*>(op) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            } else {                                               // int8 operand of the consumer (to_out / proj_out)
                int8_t* op = reinterpret_cast<int8_t*>(out) + b * sO + (int64_t)q * ldo + col;
                uint32_t pk = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = fminf(fmaxf(rint_div(v4[e], oq.d, oq.inv) + oq.z, 0.f), oq.qmax);
                    pk |= (uint32_t)(uint8_t)(int8_t)((int)c - 128) << (8 * e);
                }
                *reinterpret_cast<uint32_t*>(op) = pk;
            }
        }
}


// ---- K6w: the same attention for ONE WIDE head (class-conditional LDM-4: 1 head of d = 384 at 32 x 32, openaimodel.py:384-406) ----
// With d in the hundreds the two products are long-K GEMMs and the kernel is MFMA-bound, not VALU-bound: a wave keeps its 32
// queries' Q fragments (d / 16 x 4 registers) AND the whole O^T accumulator (d / 32 blocks x 16 registers) in the 512-entry
// register file of a one-wave-per-SIMD kernel; keys and values are staged in 32-key blocks.  K rows sit at a stride of d + 8
// halfs (= 4 banks mod 64: conflict-free ds_read_b128 of the 32x32x16 A operand), V rows stay [key][d] at a stride of d + 32 halfs
// (= 16 banks mod 64) and the P V product's A operand V^T is fetched with the gfx950 transposing read ds_read_b64_tr_b16: a
// 16-lane group reads 4 keys x 16 dimensions and every lane receives one dimension's 4 keys -- the two reads of a lane half give
// keys {4h .. 4h+3, 8+4h .. 8+4h+3} of a 16-key step, exactly the order in which the accumulator of S^T holds a lane's
// probabilities.  Same two walks, same arithmetic, same codes as k_attn_fused.
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define ATTW_BK 32
#ifdef EDADM_STAMPS              // diagnostic build (tools/attn_wide_bench.hip): cycle stamps of wave 0 of every workgroup
__device__ unsigned long long g_attw_stamps[16];
#define ATTW_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define ATTW_ADD(slot, d) attw_loc[slot] += (unsigned long long)(d)       // per-wave registers; one atomic per slot at the end
#else
#define ATTW_T(v)
#define ATTW_ADD(slot, d)
#endif

template <int KD, int DVB>
__global__ void __launch_bounds__(256)
k_attn_wide(const __half* __restrict__ Q, int64_t ldq, int64_t sQ, int64_t hQ, const __half* __restrict__ K, int64_t ldk,
            int64_t sK, int64_t hK, const __half* __restrict__ V, int64_t ldv, int64_t sV, int64_t hV, void* __restrict__ out,
            int64_t ldo, int64_t sO, int Nq, int Nk, float alpha_qk, const QP* __restrict__ pqp, float alpha_pv, int out_mode,
            const QP* __restrict__ oqp) {
    const QP pw = qp_load(pqp, 0);
#ifdef EDADM_STAMPS
    unsigned long long attw_loc[16] = {0};
#endif
    constexpr int D = KD * 16;
    static_assert(DVB * 32 == D, "d must be a multiple of 32");
    constexpr int KROW = D + 8, VROW = D + 32;
    __shared__ __half lk_[2][ATTW_BK * KROW];
    __shared__ __half lv_[2][ATTW_BK * VROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    // Workgroups are dealt round-robin to the 8 XCDs by their linear index, and every query tile of an (image, head) walks the same
    // K and V twice: XCD k takes the contiguous range [k n / 8, (k + 1) n / 8) of (image, head, tile) triples, so that an image's
    // tiles share one L2 (n % 8 != 0: plain numbering)
    int qt, h;
    int64_t b;
    {
        const unsigned nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
        const unsigned id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
        const unsigned L = (n & 7) ? id : (n >> 3) * (id & 7) + (id >> 3);
        qt = (int)(L % nx);
        h = (int)((L / nx) % ny);
        b = L / (nx * ny);
    }
    const int q = qt * ATT_BQ + wave * 32 + fr;
    const __half* Qb = Q + b * sQ + (int64_t)h * hQ;
    const __half* Kb = K + b * sK + (int64_t)h * hK;
    const __half* Vb = V + b * sV + (int64_t)h * hV;
    half8 qf[KD];
#pragma unroll
    for (int ks = 0; ks < KD; ++ks) {
        half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q < Nq) z = *reinterpret_cast<const half8*>(Qb + (int64_t)q * ldq + ks * 16 + fh * 8);
        qf[ks] = z;
    }
    const int nkb = Nk / ATTW_BK;                                  // the launcher guarantees Nk % 64 == 0
    // staging: a thread owns the 16-byte chunks (key r0 + 16 a, chunk c0 + 16 c), a = 0, 1, c = 0 .. D / 128 - 1, of every block: one
    // 32-bit offset per tensor and row half, everything else is an instruction immediate; a wave reads 4 x 256 contiguous bytes
    static_assert(D % 128 == 0, "d must be a multiple of 128");
    constexpr int CPR = D / 128;                                   // chunk columns per thread
    constexpr int KPT = 2 * CPR;
    half8 rk[KPT], rv[KPT];
    const int r0 = tid >> 4, c0 = tid & 15;
    const int kofs0 = r0 * (int)ldk + c0 * 8, kofs1 = kofs0 + 16 * (int)ldk;
    const int vofs0 = r0 * (int)ldv + c0 * 8, vofs1 = vofs0 + 16 * (int)ldv;
    const int klds = r0 * KROW + c0 * 8, vlds = r0 * VROW + c0 * 8;
    auto gload_k = [&](int kb) {
        const __half* base = Kb + (int64_t)kb * ATTW_BK * ldk;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < CPR; ++c) rk[a * CPR + c] = *reinterpret_cast<const half8*>(base + (a ? kofs1 : kofs0) + c * 128);
    };
    auto gload_v = [&](int kb) {
        const __half* base = Vb + (int64_t)kb * ATTW_BK * ldv;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < CPR; ++c) rv[a * CPR + c] = *reinterpret_cast<const half8*>(base + (a ? vofs1 : vofs0) + c * 128);
    };
    auto lstore_k = [&](int buf) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < CPR; ++c) *reinterpret_cast<half8*>(lk_[buf] + klds + a * 16 * KROW + c * 128) = rk[a * CPR + c];
    };
    auto lstore_v = [&](int buf) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < CPR; ++c) *reinterpret_cast<half8*>(lv_[buf] + vlds + a * 16 * VROW + c * 128) = rv[a * CPR + c];
    };
    // S^T of one 32-key block for this wave's 32 queries.  The fragment reads run PF steps ahead of the MFMAs that consume them
    // (left to itself hipcc issues each read right in front of its MFMA and waits out the LDS latency 24 times per block), and
    // `valu` single-issue instructions of whatever independent arithmetic the caller placed in the same region -- the softmax of
    // the PREVIOUS block -- go into every MFMA's shadow: one wave per SIMD, nothing else fills the matrix pipe's gaps
    constexpr int PF = 6;
    auto scores = [&](const __half* lk, float16v& c, auto valu) {
        c = float16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const __half* base = lk + fr * KROW + fh * 8;
        half8 a[KD];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) a[ks] = *reinterpret_cast<const half8*>(base + ks * 16);
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            if (ks + PF < KD) a[ks + PF] = *reinterpret_cast<const half8*>(base + (ks + PF) * 16);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], qf[ks], c, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int ks = 0; ks < KD; ++ks) {
            if (ks + PF < KD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (decltype(valu)::value > 0) __builtin_amdgcn_sched_group_barrier(0x2, decltype(valu)::value, 0);
        }
    };
    const float cexp = alpha_qk * 1.44269504088896340736f;

    // ---- walk 1: row maximum and row sum (online form).  Pipeline: at the top of iteration kb the scores of block kb are in
    // registers, K(kb + 1) is resident in buffer (kb + 1) & 1 and K(kb + 2) is on its way from L2 in registers
    float mx = -INFINITY, sum = 0.f;
    auto stats = [&](const float16v& acc) {
        float bm = mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) bm = fmaxf(bm, acc[r]);
        sum *= __builtin_amdgcn_exp2f((mx - bm) * cexp);
        mx = bm;
        const float cm = mx * cexp;
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) part += __builtin_amdgcn_exp2f(fmaf(acc[r], cexp, -cm));
        sum += part;
    };
    // Order inside an iteration: FIRST the LDS stores of the blocks that arrived during the previous iteration (they drain beside
    // the MFMAs: 48 KB per block is 600 LDS-array cycles at the ds_write_b128 rate), then the next blocks' global loads (a whole
    // iteration to land), then the arithmetic, then the one barrier.  Top of iteration kb: scores of block kb in registers, K(kb + 1)
    // resident in buffer (kb + 1) & 1, K(kb + 2) in the staging registers.
    float16v sa, sb;
    gload_k(0);
    lstore_k(0);
    gload_k(1);
    lstore_k(1);
    __syncthreads();
    if (nkb > 2) gload_k(2);
    scores(lk_[0], sa, std::integral_constant<int, 0>{});
    __syncthreads();                                               // every wave is done with K(0) before iteration 0 overwrites it
    auto step1 = [&](int kb, float16v& cur, float16v& nxt) {
        ATTW_T(t0);
        if (kb + 2 < nkb) lstore_k(kb & 1);
        if (kb + 3 < nkb) gload_k(kb + 3);
        ATTW_T(t1);
        if (kb + 1 < nkb) scores(lk_[(kb + 1) & 1], nxt, std::integral_constant<int, 3>{});
        stats(cur);
        ATTW_T(t2);
        __syncthreads();
        ATTW_T(t3);
        ATTW_ADD(0, t1 - t0); ATTW_ADD(1, t2 - t1); ATTW_ADD(2, t3 - t2); ATTW_ADD(5, 1);
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        step1(kb, sa, sb);
        step1(kb + 1, sb, sa);
    }
    {
        const float mo = __shfl_xor(mx, 32, 64), so = __shfl_xor(sum, 32, 64);
        const float m = fmaxf(mx, mo);
        sum = sum * __builtin_amdgcn_exp2f((mx - m) * cexp) + so * __builtin_amdgcn_exp2f((mo - m) * cexp);
        mx = m;
    }
    const float cmax = mx * cexp;
    const float inv = 1.0f / (sum * pw.d);
    const bool z0 = pw.z == 0.f;
    float16v o[DVB];
#pragma unroll
    for (int j = 0; j < DVB; ++j) o[j] = float16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // transposing read: lane 4 q4 + p of a 16-lane group supplies the address of key q4, dimensions 4 p .. 4 p + 3 of the group's 16
    const int tr_off = (4 * fh + ((lane & 15) >> 2)) * VROW + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    // ---- walk 2: probability codes and O^T = V^T P^T.  Same K pipeline; V(kb) is resident in buffer kb & 1 at the top of
    // iteration kb and V(kb + 1) on its way.  The exponentials and roundings of block kb sit in the shadow of the MFMAs that
    // compute the scores of block kb + 1.
    auto pv = [&](const __half* lv, const half8 (&pf)[2]) {
        // 2 x DVB steps (16-key step j2, 32-dimension block j), the two transposing reads of a step PF steps ahead of its MFMA
        constexpr int NS = 2 * DVB;
        const __half* vp = lv + tr_off;
        half4 va[NS], vb[NS];
        auto rd = [&](int st) {
            const int j2 = st / DVB, j = st - j2 * DVB;
            va[st] = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4*)(vp + j2 * 16 * VROW + j * 32)));
            vb[st] = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4*)(vp + j2 * 16 * VROW + j * 32 + 8 * VROW)));
        };
#pragma unroll
        for (int st = 0; st < PF; ++st) rd(st);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + PF < NS) rd(st + PF);
            const int j2 = st / DVB, j = st - j2 * DVB;
            const half8 a = {va[st][0], va[st][1], va[st][2], va[st][3], vb[st][0], vb[st][1], vb[st][2], vb[st][3]};
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pf[j2], o[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 1);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + PF < NS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 1);
        }
    };
    gload_k(0);
    gload_v(0);
    lstore_k(0);
    lstore_v(0);
    gload_k(1);
    lstore_k(1);
    __syncthreads();
    if (nkb > 2) gload_k(2);
    gload_v(1);
    scores(lk_[0], sa, std::integral_constant<int, 0>{});
    __syncthreads();
    auto step2 = [&](int kb, float16v& cur, float16v& nxt) {
        ATTW_T(t0);
        if (kb + 2 < nkb) lstore_k(kb & 1);
        if (kb + 1 < nkb) lstore_v((kb + 1) & 1);
        if (kb + 3 < nkb) gload_k(kb + 3);
        if (kb + 2 < nkb) gload_v(kb + 2);
        ATTW_T(t1);
        float r_[16];
        float worst = 0.f;
        if (kb + 1 < nkb) scores(lk_[(kb + 1) & 1], nxt, std::integral_constant<int, 4>{});
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(cur[r], cexp, -cmax));
            cur[r] = e;
            const float t = e * inv;
            r_[r] = rintf(t);
            worst = fmaxf(worst, fmaf(t, EDADM_SM_BAND_REL, fabsf(t - r_[r])));
        }
        ATTW_T(t2);
        if (__builtin_expect(worst > EDADM_SM_NEAR, 0)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asm volatile("" : "+v"(r_[r]));
                r_[r] = rintf((cur[r] / sum) / pw.d);
            }
        }
        half8 pf[2];
        if (z0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pf[r >> 3][r & 7] = (_Float16)fminf(r_[r], pw.qmax);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) pf[r >> 3][r & 7] = (_Float16)(fminf(fmaxf(r_[r] + pw.z, 0.f), pw.qmax) - pw.z);
        }
        pv(lv_[kb & 1], pf);
        ATTW_T(t3);
        __syncthreads();
        ATTW_T(t4);
        ATTW_ADD(8, t1 - t0); ATTW_ADD(9, t2 - t1); ATTW_ADD(10, t3 - t2); ATTW_ADD(11, t4 - t3); ATTW_ADD(13, 1);
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        step2(kb, sa, sb);
        step2(kb + 1, sb, sa);
    }
#ifdef EDADM_STAMPS
    if (tid == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_attw_stamps[i], attw_loc[i]);
#endif
    if (q >= Nq) return;
    QP oq;
    if (out_mode == 2) oq = qp_load(oqp, 0);
#pragma unroll
    for (int j = 0; j < DVB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dv = j * 32 + 8 * g + 4 * fh;
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = o[j][4 * g + e] * alpha_pv;
            const int64_t col = (int64_t)h * D + dv;
            if (out_mode == 0) {
                float* op = reinterpret_cast<float*>(out) + b * sO + (int64_t)q * ldo + col;
                *reinterpret_cast<float4*>(op) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            } else {
                int8_t* op = reinterpret_cast<int8_t*>(out) + b * sO + (int64_t)q * ldo + col;
                uint32_t pk = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = fminf(fmaxf(rint_div(v4[e], oq.d, oq.inv) + oq.z, 0.f), oq.qmax);
                    pk |= (uint32_t)(uint8_t)(int8_t)((int)c - 128) << (8 * e);
                }
                *reinterpret_cast<uint32_t*>(op) = pk;
            }
        }
}

// ---- K6w, two waves per SIMD.  k_attn_wide above keeps a wave's whole state in a 512-register file and therefore runs ONE wave per
// SIMD: whatever the wave issues besides MFMAs (a quarter of its instructions) or waits for (LDS latency, the block barrier)
// leaves the matrix pipe idle -- 31 % MFMA utilisation measured (SQ_VALU_MFMA_BUSY_CYCLES / wave cycles; LDS array 17 % busy, no bank
// conflicts).  This form gives a wave 16 queries on the 16x16x32 MFMA: Q fragments d / 32 x 4 = 48 registers, O^T d / 16 x 4 = 96,
// so eight waves of 16 queries (the same 128-query tile, the same K / V staging per workgroup) fit two to a SIMD and one wave's
// exponentials and waits sit under the other's MFMAs.  In the 16x16 accumulator layout a lane holds query (lane & 15) and keys
// 4 g + r (g = lane >> 4) of a 16-key tile; the 8 values of two tiles, as f16, are the B operand of ONE 32-key P V step when V^T is
// read in the key order {4g .. 4g+3, 16+4g .. 16+4g+3}: two transposing reads per 16-dimension block.  K and V rows both sit at a
// stride of d + 16 halfs (8 banks mod 64): conflict-free for the 16-row ds_read_b128 fragments and for the transposing reads.
typedef float float4v __attribute__((ext_vector_type(4)));
#ifdef EDADM_DIAG
__device__ int g_attw_exp;
#endif

template <int KD32, int DB16>
__global__ void __launch_bounds__(512)
k_attn_wide16(const __half* __restrict__ Q, int64_t ldq, int64_t sQ, int64_t hQ, const __half* __restrict__ K, int64_t ldk,
              int64_t sK, int64_t hK, const __half* __restrict__ V, int64_t ldv, int64_t sV, int64_t hV, void* __restrict__ out,
              int64_t ldo, int64_t sO, int Nq, int Nk, float alpha_qk, const QP* __restrict__ pqp, float alpha_pv, int out_mode,
              const QP* __restrict__ oqp) {
    const QP pw = qp_load(pqp, 0);
#ifdef EDADM_DIAG
    const int exp_ = g_attw_exp;                                   // timing experiments (wrong results): 1 no block barrier, 2 no LDS stores, 4 no P V
#define ATTW_SYNC() do { if (!(exp_ & 1)) __syncthreads(); } while (0)
#define ATTW_EXP(bit) (exp_ & (bit))
#else
#define ATTW_SYNC() __syncthreads()
#define ATTW_EXP(bit) 0
#endif
    constexpr int D = KD32 * 32;
    static_assert(DB16 * 16 == D && D % 128 == 0, "d must be a multiple of 128");
    constexpr int ROW = D + 16;                                    // halfs per LDS row of K and of V
    __shared__ __half lk_[2][ATTW_BK * ROW];
    __shared__ __half lv_[2][ATTW_BK * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fc = lane & 15, fg = lane >> 4;
    int qt, h;
    int64_t b;
    {   // XCD k takes a contiguous range of (image, head, tile) triples: an image's tiles share one L2 (see k_attn_wide)
        const unsigned nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
        const unsigned id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
        const unsigned L = (n & 7) ? id : (n >> 3) * (id & 7) + (id >> 3);
        qt = (int)(L % nx);
        h = (int)((L / nx) % ny);
        b = L / (nx * ny);
    }
    const int q = qt * ATT_BQ + wave * 16 + fc;
    const __half* Qb = Q + b * sQ + (int64_t)h * hQ;
    const __half* Kb = K + b * sK + (int64_t)h * hK;
    const __half* Vb = V + b * sV + (int64_t)h * hV;
    half8 qf[KD32];                                                // B operand of S^T = K Q^T: query fc, dimensions 32 ks + 8 fg .. + 7
#pragma unroll
    for (int ks = 0; ks < KD32; ++ks) {
        half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q < Nq) z = *reinterpret_cast<const half8*>(Qb + (int64_t)q * ldq + ks * 32 + fg * 8);
        qf[ks] = z;
    }
    const int nkb = Nk / ATTW_BK;                                  // the launcher guarantees Nk % 64 == 0
    // staging: thread -> key r0 = tid >> 4, 16-byte chunks c0 + 16 c (c < D / 128) of every block
    constexpr int CPR = D / 128;
    half8 rk[CPR], rv[CPR];
    const int r0 = tid >> 4, c0 = tid & 15;
    const int kofs = r0 * (int)ldk + c0 * 8, vofs = r0 * (int)ldv + c0 * 8, sofs = r0 * ROW + c0 * 8;
    auto gload_k = [&](int kb) {
        const __half* base = Kb + (int64_t)kb * ATTW_BK * ldk;
#pragma unroll
        for (int c = 0; c < CPR; ++c) rk[c] = *reinterpret_cast<const half8*>(base + kofs + c * 128);
    };
    auto gload_v = [&](int kb) {
        const __half* base = Vb + (int64_t)kb * ATTW_BK * ldv;
#pragma unroll
        for (int c = 0; c < CPR; ++c) rv[c] = *reinterpret_cast<const half8*>(base + vofs + c * 128);
    };
    auto lstore_k = [&](int buf) {
        if (ATTW_EXP(2)) return;
#pragma unroll
        for (int c = 0; c < CPR; ++c) *reinterpret_cast<half8*>(lk_[buf] + sofs + c * 128) = rk[c];
    };
    auto lstore_v = [&](int buf) {
        if (ATTW_EXP(2)) return;
#pragma unroll
        for (int c = 0; c < CPR; ++c) *reinterpret_cast<half8*>(lv_[buf] + sofs + c * 128) = rv[c];
    };
    // S^T of a 32-key block: two 16-key tiles x this wave's 16 queries; register r of tile t is key 16 t + 4 fg + r
    constexpr int PF = 4;
    auto scores = [&](const __half* lk, float4v (&c)[2]) {
        c[0] = c[1] = float4v{0, 0, 0, 0};
        const __half* base = lk + fc * ROW + fg * 8;
        constexpr int NS = 2 * KD32;
        half8 a[NS];
        auto rd = [&](int st) { a[st] = *reinterpret_cast<const half8*>(base + (st / KD32) * 16 * ROW + (st % KD32) * 32); };
#pragma unroll
        for (int st = 0; st < PF; ++st) rd(st);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + PF < NS) rd(st + PF);
            c[st / KD32] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[st], qf[st % KD32], c[st / KD32], 0, 0, 0);
        }
    };
    const float cexp = alpha_qk * 1.44269504088896340736f;
    float mx = -INFINITY, sum = 0.f;
    auto stats = [&](const float4v (&acc)[2]) {
        float bm = mx;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) bm = fmaxf(bm, acc[t][r]);
        sum *= __builtin_amdgcn_exp2f((mx - bm) * cexp);
        mx = bm;
        const float cm = mx * cexp;
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) part += __builtin_amdgcn_exp2f(fmaf(acc[t][r], cexp, -cm));
        sum += part;
    };
    // ---- walk 1 (pipeline as in k_attn_wide: stores of the arrived block, next loads, scores of block kb + 1 beside the statistics of kb)
    float4v sa[2], sb[2];
    gload_k(0);
    lstore_k(0);
    gload_k(1);
    lstore_k(1);
    __syncthreads();
    if (nkb > 2) gload_k(2);
    scores(lk_[0], sa);
    __syncthreads();
    auto step1 = [&](int kb, float4v (&cur)[2], float4v (&nxt)[2]) {
        if (kb + 2 < nkb) lstore_k(kb & 1);
        if (kb + 3 < nkb) gload_k(kb + 3);
        if (kb + 1 < nkb) scores(lk_[(kb + 1) & 1], nxt);
        stats(cur);
        ATTW_SYNC();
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        step1(kb, sa, sb);
        step1(kb + 1, sb, sa);
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {                        // the four lanes of a query (4 keys of every 16 each) combine
        const float mo = __shfl_xor(mx, sh, 64), so = __shfl_xor(sum, sh, 64);
        const float m = fmaxf(mx, mo);
        sum = sum * __builtin_amdgcn_exp2f((mx - m) * cexp) + so * __builtin_amdgcn_exp2f((mo - m) * cexp);
        mx = m;
    }
    const float cmax = mx * cexp;
    const float inv = 1.0f / (sum * pw.d);
    const bool z0 = pw.z == 0.f;
    float4v o[DB16];
#pragma unroll
    for (int j = 0; j < DB16; ++j) o[j] = float4v{0, 0, 0, 0};
    // transposing read: lane 4 q4 + p of a 16-lane group supplies the address of key 4 fg + q4, dimensions 4 p .. 4 p + 3
    const int tr_off = (4 * fg + ((lane & 15) >> 2)) * ROW + 4 * (lane & 3);
    auto pv = [&](const __half* lv, const half8& pf) {
        const __half* vp = lv + tr_off;
        half4 va[DB16], vb[DB16];
        auto rd = [&](int j) {
            va[j] = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(vp + j * 16)));
            vb[j] = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4*)(vp + j * 16 + 16 * ROW)));
        };
#pragma unroll
        for (int j = 0; j < PF; ++j) rd(j);
#pragma unroll
        for (int j = 0; j < DB16; ++j) {
            if (j + PF < DB16) rd(j + PF);
            const half8 a = {va[j][0], va[j][1], va[j][2], va[j][3], vb[j][0], vb[j][1], vb[j][2], vb[j][3]};
            o[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf, o[j], 0, 0, 0);
        }
    };
    gload_k(0);
    gload_v(0);
    lstore_k(0);
    lstore_v(0);
    gload_k(1);
    lstore_k(1);
    __syncthreads();
    if (nkb > 2) gload_k(2);
    gload_v(1);
    scores(lk_[0], sa);
    __syncthreads();
    auto step2 = [&](int kb, float4v (&cur)[2], float4v (&nxt)[2]) {
        if (kb + 2 < nkb) lstore_k(kb & 1);
        if (kb + 1 < nkb) lstore_v((kb + 1) & 1);
        if (kb + 3 < nkb) gload_k(kb + 3);
        if (kb + 2 < nkb) gload_v(kb + 2);
        if (kb + 1 < nkb) scores(lk_[(kb + 1) & 1], nxt);
        float e_[8], r_[8];
        float worst = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float e = __builtin_amdgcn_exp2f(fmaf(cur[i >> 2][i & 3], cexp, -cmax));
            e_[i] = e;
            const float t = e * inv;
            r_[i] = rintf(t);
            worst = fmaxf(worst, fmaf(t, EDADM_SM_BAND_REL, fabsf(t - r_[i])));
        }
        if (__builtin_expect(worst > EDADM_SM_NEAR, 0)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("" : "+v"(r_[i]));
                r_[i] = rintf((e_[i] / sum) / pw.d);
            }
        }
        half8 pf;
        if (z0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) pf[i] = (_Float16)fminf(r_[i], pw.qmax);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) pf[i] = (_Float16)(fminf(fmaxf(r_[i] + pw.z, 0.f), pw.qmax) - pw.z);
        }
        if (!ATTW_EXP(4)) pv(lv_[kb & 1], pf);
        else if (kb == -1) pv(lv_[kb & 1], pf);
        else o[0][0] += (float)pf[0] + (float)pf[7];
        ATTW_SYNC();
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        step2(kb, sa, sb);
        step2(kb + 1, sb, sa);
    }
    if (q >= Nq) return;
    // ---- epilogue: lane -> query q, output dimensions 16 j + 4 fg + e
    QP oq;
    if (out_mode == 2) oq = qp_load(oqp, 0);
#pragma unroll
    for (int j = 0; j < DB16; ++j) {
        const int dv = j * 16 + 4 * fg;
        float v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = o[j][e] * alpha_pv;
        const int64_t col = (int64_t)h * D + dv;
        if (out_mode == 0) {
            float* op = reinterpret_cast<float*>(out) + b * sO + (int64_t)q * ldo + col;
            *reinterpret_cast<float4*>(op) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        } else {
            int8_t* op = reinterpret_cast<int8_t*>(out) + b * sO + (int64_t)q * ldo + col;
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = fminf(fmaxf(rint_div(v4[e], oq.d, oq.inv) + oq.z, 0.f), oq.qmax);
                pk |= (uint32_t)(uint8_t)(int8_t)((int)c - 128) << (8 * e);
            }
            *reinterpret_cast<uint32_t*>(op) = pk;
        }
    }
}

// ---- K6s: one wide head over FEW keys (LDM-4's 16 x 16 level: d = 576, 256 keys; 8 x 8: d = 960, 64 keys).  All scores of a
// query fit a lane's registers (Nk / 4 values: a lane holds keys 4 g + r of every 16-key tile), so the keys are walked ONCE:
// S tiles kept, row maximum / sum / probability codes in registers, then V walked once -- no recomputation, one launch instead of
// three.  Same 16-query waves, operand layouts, staging and arithmetic as k_attn_wide16; four waves (64 queries) per workgroup so
// that the small levels still give every CU work (400 workgroups at 16 x 16, 100 at 8 x 8).  When the O^T accumulator of the whole
// head does not fit beside the codes (d = 960: 240 registers), the output dimensions are processed in DH passes over V.
template <int KD32, int NT /* 16-key tiles */, int DH>
__global__ void __launch_bounds__(256)
k_attn_small(const __half* __restrict__ Q, int64_t ldq, int64_t sQ, int64_t hQ, const __half* __restrict__ K, int64_t ldk,
             int64_t sK, int64_t hK, const __half* __restrict__ V, int64_t ldv, int64_t sV, int64_t hV, void* __restrict__ out,
             int64_t ldo, int64_t sO, int Nq, int Nk, float alpha_qk, const QP* __restrict__ pqp, float alpha_pv, int out_mode,
             const QP* __restrict__ oqp) {
    const QP pw = qp_load(pqp, 0);
    constexpr int D = KD32 * 32, ROW = D + 16;
    constexpr int DB = D / 16 / DH;                                // 16-dimension blocks per pass
    static_assert(D % 64 == 0 && (D / 16) % DH == 0 && NT % 2 == 0, "shape");
    __shared__ __half lb_[2][ATTW_BK * ROW];                        // K blocks, then V blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fc = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y;
    const int64_t b = blockIdx.z;
    const int q = blockIdx.x * 64 + wave * 16 + fc;
    const __half* Qb = Q + b * sQ + (int64_t)h * hQ;
    const __half* Kb = K + b * sK + (int64_t)h * hK;
    const __half* Vb = V + b * sV + (int64_t)h * hV;
    constexpr int NB = NT / 2;                                     // 32-key blocks (the launcher guarantees Nk == 16 NT)
    constexpr int CPR = D / 64;                                    // staging: thread -> key r0 = tid / 8, 16-byte chunks c0 + 8 c of its row
    const int r0 = tid >> 3, c0 = tid & 7;
    // two staging register sets: blocks kb + 1 and kb + 2 are on their way from L2 while block kb computes (an iteration is a few
    // hundred cycles of MFMA: with one block in flight every iteration waited out the whole load latency)
    half8 rs[2][CPR];
    auto gload = [&](const __half* base, int64_t ld, int kb) {
        const __half* p = base + ((int64_t)kb * ATTW_BK + r0) * ld + c0 * 8;
#pragma unroll
        for (int c = 0; c < CPR; ++c) rs[kb & 1][c] = *reinterpret_cast<const half8*>(p + c * 64);
    };
    auto lstore = [&](int kb) {
#pragma unroll
        for (int c = 0; c < CPR; ++c) *reinterpret_cast<half8*>(lb_[kb & 1] + r0 * ROW + c0 * 8 + c * 64) = rs[kb & 1][c];
    };
    float4v s[NT];
    {
        half8 qf[KD32];
#pragma unroll
        for (int ks = 0; ks < KD32; ++ks) {
            half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            if (q < Nq) z = *reinterpret_cast<const half8*>(Qb + (int64_t)q * ldq + ks * 32 + fg * 8);
            qf[ks] = z;
        }
        gload(Kb, ldk, 0);
        if (NB > 1) gload(Kb, ldk, 1);
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            lstore(kb);
            __syncthreads();
            if (kb + 2 < NB) gload(Kb, ldk, kb + 2);
            const __half* base = lb_[kb & 1] + fc * ROW + fg * 8;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float4v c = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KD32; ++ks) {
                    const half8 a = *reinterpret_cast<const half8*>(base + t * 16 * ROW + ks * 32);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[ks], c, 0, 0, 0);
                }
                s[2 * kb + t] = c;
            }
        }
    }
    // ---- softmax in registers: the four lanes of a query (4 keys of every 16 each) combine their maxima and sums
    const float cexp = alpha_qk * 1.44269504088896340736f;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float cmax = mx * cexp;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], cexp, -cmax));
            sum += s[t][r];
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / (sum * pw.d);
    const bool z0 = pw.z == 0.f;
    half8 pf[NB];                                                  // block kb: keys {4g .. 4g+3} of tile 2 kb, then of tile 2 kb + 1
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        float r_[8];
        float worst = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = s[2 * kb + (i >> 2)][i & 3] * inv;
            r_[i] = rintf(t);
            worst = fmaxf(worst, fmaf(t, EDADM_SM_BAND_REL, fabsf(t - r_[i])));
        }
        if (__builtin_expect(worst > EDADM_SM_NEAR, 0)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("" : "+v"(r_[i]));
                r_[i] = rintf((s[2 * kb + (i >> 2)][i & 3] / sum) / pw.d);
            }
        }
        if (z0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) pf[kb][i] = (_Float16)fminf(r_[i], pw.qmax);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) pf[kb][i] = (_Float16)(fminf(fmaxf(r_[i] + pw.z, 0.f), pw.qmax) - pw.z);
        }
    }
    // ---- O^T = V^T P^T, DH passes over V of D / DH output dimensions each
    const int tr_off = (4 * fg + ((lane & 15) >> 2)) * ROW + 4 * (lane & 3);
    QP oq;
    if (out_mode == 2) oq = qp_load(oqp, 0);
#pragma unroll 1
    for (int dh = 0; dh < DH; ++dh) {
        float4v o[DB];
#pragma unroll
        for (int j = 0; j < DB; ++j) o[j] = float4v{0, 0, 0, 0};
        __syncthreads();                                           // every wave is done with the buffers of the previous phase
        gload(Vb, ldv, 0);
        if (NB > 1) gload(Vb, ldv, 1);
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            lstore(kb);
            __syncthreads();
            if (kb + 2 < NB) gload(Vb, ldv, kb + 2);
            const __half* vp = lb_[kb & 1] + tr_off + dh * DB * 16;
#pragma unroll
            for (int j = 0; j < DB; ++j) {
                const half4 va = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4*)(vp + j * 16)));
                const half4 vb = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4*)(vp + j * 16 + 16 * ROW)));
                const half8 a = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                o[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[kb], o[j], 0, 0, 0);
            }
        }
        if (q < Nq) {
#pragma unroll
            for (int j = 0; j < DB; ++j) {
                const int dv = (dh * DB + j) * 16 + 4 * fg;
                float v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = o[j][e] * alpha_pv;
                const int64_t col = (int64_t)h * D + dv;
                if (out_mode == 0) {
                    float* op = reinterpret_cast<float*>(out) + b * sO + (int64_t)q * ldo + col;
                    *reinterpret_cast<float4*>(op) = make_float4(v4[0], v4[1], v4[2], v4[3]);
                } else {
                    int8_t* op = reinterpret_cast<int8_t*>(out) + b * sO + (int64_t)q * ldo + col;
                    uint32_t pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float c = fminf(fmaxf(rint_div(v4[e], oq.d, oq.inv) + oq.z, 0.f), oq.qmax);
                        pk |= (uint32_t)(uint8_t)(int8_t)((int)c - 128) << (8 * e);
                    }
                    *reinterpret_cast<uint32_t*>(op) = pk;
                }
            }
        }
    }
}
// shapes of the few-key kernel
static bool attn_small_shape(int64_t d, int64_t Nq, int64_t Nk) { return (d == 576 && Nk == 256) || (d == 960 && Nk == 64); }

// ---- K6w with the score product on the INT8 MFMA.  The two score walks of k_attn_wide16 are bound by LDS reads (a 16x16x32 f16 MFMA
// consumes a 1 KB K fragment every 16 cycles: four SIMDs ask for the LDS array's whole 256 B / clk).  q and k are 8-bit codes: as
// int8 operands (code - 128, the projection epilogues' out_mode 2) a 16x16x64 MFMA takes the same 1 KB fragment for TWICE the keys x
// dimensions, the staged K block is half the bytes, and the sum is exact in int32 at any d.  With q8 = code_q - 128 = (code_q - z_q) -
// c_q, c_q = 128 - z_q (likewise k):  (q8 + c_q) . (k8 + c_k) = q8 . k8 + c_q sum_d k8 + [c_k sum_d q8 + d c_q c_k]; the bracket does not
// depend on the key and cancels in the softmax of a query, so the kernel adds c_q x ksum[key] only (ksum from the staging pass:
// v_dot4 with ones, reduced over the 8 lanes of a row) and exponentiates (s - max) c, a difference of exact integers.  P V stays on
// the f16 MFMA (probability codes 0 .. 255 do not fit int8; V through the transposing read as before).  64 keys per iteration.
typedef int v4i_t __attribute__((ext_vector_type(4)));

template <int KD64, int DB16>
__global__ void __launch_bounds__(512)
k_attn_wide16_i8(const int8_t* __restrict__ Q, int64_t ldq, int64_t sQ, int64_t hQ, const int8_t* __restrict__ K, int64_t ldk,
                 int64_t sK, int64_t hK, const __half* __restrict__ V, int64_t ldv, int64_t sV, int64_t hV, void* __restrict__ out,
                 int64_t ldo, int64_t sO, int Nq, int Nk, float alpha_qk, float cq, const QP* __restrict__ pqp, float alpha_pv,
                 int out_mode, const QP* __restrict__ oqp) {
    const QP pw = qp_load(pqp, 0);
    constexpr int D = KD64 * 64;
    static_assert(DB16 * 16 == D && D % 128 == 0, "d must be a multiple of 128");
    constexpr int KROWB = D + 32;                                  // bytes per LDS row of K (int8): 8 banks mod 64 beyond a multiple of 256
    constexpr int VROW = D + 16;                                   // halfs per LDS row of V
    constexpr int BK = 64;
    __shared__ __attribute__((aligned(16))) int8_t lk_[2][BK * KROWB];
    __shared__ __half lv_[2][2][32 * VROW];                         // [buffer][32-key half of the block]
    __shared__ float ksum_[2][BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fc = lane & 15, fg = lane >> 4;
    int qt, h;
    int64_t b;
    {
        const unsigned nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
        const unsigned id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
        const unsigned L = (n & 7) ? id : (n >> 3) * (id & 7) + (id >> 3);
        qt = (int)(L % nx);
        h = (int)((L / nx) % ny);
        b = L / (nx * ny);
    }
    const int q = qt * ATT_BQ + wave * 16 + fc;
    const int8_t* Qb = Q + b * sQ + (int64_t)h * hQ;
    const int8_t* Kb = K + b * sK + (int64_t)h * hK;
    const __half* Vb = V + b * sV + (int64_t)h * hV;
    v4i_t qf[KD64];                                                // B operand: query fc, dimensions 64 ks + 16 fg .. + 15
#pragma unroll
    for (int ks = 0; ks < KD64; ++ks) {
        v4i_t z = {0, 0, 0, 0};
        if (q < Nq) z = *reinterpret_cast<const v4i_t*>(Qb + (int64_t)q * ldq + ks * 64 + fg * 16);
        qf[ks] = z;
    }
    const int nkb = Nk / BK;                                       // the launcher guarantees Nk % 64 == 0
    // staging.  K: thread -> key tid / 8, 16-byte chunks (tid & 7) + 8 c of its 384-byte row.  V: thread -> key tid / 16 of each
    // 32-key half, chunks (tid & 15) + 16 c
    constexpr int KC = D / 128, VC = D / 128;
    v4i_t rk[KC];
    half8 rv[2][VC];
    const int kr = tid >> 3, kc0 = tid & 7, vr = tid >> 4, vc0 = tid & 15;
    auto gload_k = [&](int kb) {
        const int8_t* p = Kb + ((int64_t)kb * BK + kr) * ldk + kc0 * 16;
#pragma unroll
        for (int c = 0; c < KC; ++c) rk[c] = *reinterpret_cast<const v4i_t*>(p + c * 128);
    };
    auto gload_v = [&](int kb) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const __half* p = Vb + ((int64_t)kb * BK + hf * 32 + vr) * ldv + vc0 * 8;
#pragma unroll
            for (int c = 0; c < VC; ++c) rv[hf][c] = *reinterpret_cast<const half8*>(p + c * 128);
        }
    };
    auto lstore_k = [&](int buf) {
        int part = 0;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            *reinterpret_cast<v4i_t*>(lk_[buf] + kr * KROWB + kc0 * 16 + c * 128) = rk[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) part = __builtin_amdgcn_sdot4(rk[c][e], 0x01010101, part, false);
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (kc0 == 0) ksum_[buf][kr] = (float)part;
    };
    auto lstore_v = [&](int buf) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int c = 0; c < VC; ++c) *reinterpret_cast<half8*>(lv_[buf][hf] + vr * VROW + vc0 * 8 + c * 128) = rv[hf][c];
    };
    // S^T of a 64-key block: four 16-key tiles x this wave's 16 queries; register r of tile t is key 16 t + 4 fg + r
    auto scores = [&](int buf, float (&s)[16]) {
        const int8_t* base = lk_[buf] + fc * KROWB + fg * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            v4i_t c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KD64; ++ks) {
                const v4i_t a = *reinterpret_cast<const v4i_t*>(base + t * 16 * KROWB + ks * 64);
                c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, qf[ks], c, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s[4 * t + r] = fmaf(cq, ksum_[buf][16 * t + 4 * fg + r], (float)c[r]);
        }
    };
    const float cexp = alpha_qk * 1.44269504088896340736f;
    // ---- walk 1: row maximum and sum (online form)
    float mx = -INFINITY, sum = 0.f;
    gload_k(0);
    for (int kb = 0; kb < nkb; ++kb) {
        lstore_k(kb & 1);
        __syncthreads();
        if (kb + 1 < nkb) gload_k(kb + 1);
        float s[16];
        scores(kb & 1, s);
        float bm = mx;
#pragma unroll
        for (int i = 0; i < 16; ++i) bm = fmaxf(bm, s[i]);
        sum *= __builtin_amdgcn_exp2f((mx - bm) * cexp);
        mx = bm;
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) part += __builtin_amdgcn_exp2f((s[i] - mx) * cexp);
        sum += part;
    }
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        const float mo = __shfl_xor(mx, sh, 64), so = __shfl_xor(sum, sh, 64);
        const float m = fmaxf(mx, mo);
        sum = sum * __builtin_amdgcn_exp2f((mx - m) * cexp) + so * __builtin_amdgcn_exp2f((mo - m) * cexp);
        mx = m;
    }
    const float inv = 1.0f / (sum * pw.d);
    const bool z0 = pw.z == 0.f;
    float4v o[DB16];
#pragma unroll
    for (int j = 0; j < DB16; ++j) o[j] = float4v{0, 0, 0, 0};
    const int tr_off = (4 * fg + ((lane & 15) >> 2)) * VROW + 4 * (lane & 3);
    // ---- walk 2: codes and O^T = V^T P^T
    __syncthreads();                                               // every wave is done with the last K block of walk 1
    gload_k(0);
    gload_v(0);
    for (int kb = 0; kb < nkb; ++kb) {
        lstore_k(kb & 1);
        lstore_v(kb & 1);
        __syncthreads();
        if (kb + 1 < nkb) {
            gload_k(kb + 1);
            gload_v(kb + 1);
        }
        float s[16];
        scores(kb & 1, s);
        float r_[16];
        float worst = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = __builtin_amdgcn_exp2f((s[i] - mx) * cexp);
            const float t = s[i] * inv;
            r_[i] = rintf(t);
            worst = fmaxf(worst, fmaf(t, EDADM_SM_BAND_REL, fabsf(t - r_[i])));
        }
        if (__builtin_expect(worst > EDADM_SM_NEAR, 0)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("" : "+v"(r_[i]));
                r_[i] = rintf((s[i] / sum) / pw.d);
            }
        }
        half8 pf[2];                                               // half hf: tiles 2 hf, 2 hf + 1 = keys {4g .. 4g+3, 16+4g .. 16+4g+3} of its 32
        if (z0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) pf[i >> 3][i & 7] = (_Float16)fminf(r_[i], pw.qmax);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) pf[i >> 3][i & 7] = (_Float16)(fminf(fmaxf(r_[i] + pw.z, 0.f), pw.qmax) - pw.z);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const __half* vp = lv_[kb & 1][hf] + tr_off;
#pragma unroll
            for (int j = 0; j < DB16; ++j) {
                const half4 va = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4*)(vp + j * 16)));
                const half4 vb = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4*)(vp + j * 16 + 16 * VROW)));
                const half8 a = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                o[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pf[hf], o[j], 0, 0, 0);
            }
        }
    }
    if (q >= Nq) return;
    QP oq;
    if (out_mode == 2) oq = qp_load(oqp, 0);
#pragma unroll
    for (int j = 0; j < DB16; ++j) {
        const int dv = j * 16 + 4 * fg;
        float v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = o[j][e] * alpha_pv;
        const int64_t col = (int64_t)h * D + dv;
        if (out_mode == 0) {
            float* op = reinterpret_cast<float*>(out) + b * sO + (int64_t)q * ldo + col;
            *reinterpret_cast<float4*>(op) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        } else {
            int8_t* op = reinterpret_cast<int8_t*>(out) + b * sO + (int64_t)q * ldo + col;
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = fminf(fmaxf(rint_div(v4[e], oq.d, oq.inv) + oq.z, 0.f), oq.qmax);
                pk |= (uint32_t)(uint8_t)(int8_t)((int)c - 128) << (8 * e);
            }
            *reinterpret_cast<uint32_t*>(op) = pk;
        }
    }
}

extern "C" int edadm_attention_fused_i8qk_ok(int64_t heads, int64_t d, int64_t Nq, int64_t Nk) {
    return heads >= 1 && d == 384 && Nq >= 1 && Nk >= 64 && Nk % 64 == 0;
}
extern "C" int edadm_attention_fused_i8qk(const int8_t* Q, int64_t ldq, int64_t strideQ, int64_t headQ, const int8_t* K, int64_t ldk,
                                          int64_t strideK, int64_t headK, const void* V, int64_t ldv, int64_t strideV, int64_t headV,
                                          void* out, int64_t ldo, int64_t strideO, int64_t B, int64_t heads, int64_t Nq, int64_t Nk,
                                          int64_t d, float alpha_qk, float zq, const float* pqp, float alpha_pv, int out_mode,
                                          const float* oqp, void* stream) {
    if (!Q || !K || !V || !out || !pqp || B <= 0 || !edadm_attention_fused_i8qk_ok(heads, d, Nq, Nk)) return EDADM_EINVAL;
    if ((out_mode != 0 && out_mode != 2) || (out_mode == 2 && !oqp)) return EDADM_EINVAL;
    if ((ldq & 15) || (ldk & 15) || (ldv & 7) || (strideQ & 15) || (strideK & 15) || (strideV & 7) || (headQ & 15) || (headK & 15) ||
        (headV & 7) || headQ < d || headK < d || headV < d)
        return EDADM_EINVAL;
    if (((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)V & 15)) return EDADM_EINVAL;
    if (out_mode == 0 && (((uintptr_t)out & 15) || (ldo & 3) || (strideO & 3))) return EDADM_EINVAL;
    if (out_mode == 2 && (((uintptr_t)out & 3) || (ldo & 3) || (strideO & 3))) return EDADM_EINVAL;
    if (!(alpha_qk > 0.f)) return EDADM_EINVAL;
    hipLaunchKernelGGL((k_attn_wide16_i8<6, 24>), dim3((unsigned)((Nq + ATT_BQ - 1) / ATT_BQ), (unsigned)heads, (unsigned)B), dim3(512), 0,
                       (hipStream_t)stream, Q, ldq, strideQ, headQ, K, ldk, strideK, headK, (const __half*)V, ldv, strideV, headV, out, ldo,
                       strideO, (int)Nq, (int)Nk, alpha_qk, 128.0f - zq, reinterpret_cast<const QP*>(pqp), alpha_pv, out_mode,
                       reinterpret_cast<const QP*>(oqp));
    return edadm_launch_status();
}

// shapes of the wide-head kernel: one instantiation per head dimension
static bool attn_wide_shape(int64_t d, int64_t Nq, int64_t Nk) { return d == 384 && Nk >= 64 && Nk % (2 * ATTW_BK) == 0 && Nq >= 1; }

extern "C" int edadm_attention_fused_ok(int64_t heads, int64_t d, int64_t Nq, int64_t Nk) {
    return heads >= 1 && Nq >= 1 && ((d >= 8 && d <= 160 && (d & 7) == 0 && Nk >= 2) || attn_wide_shape(d, Nq, Nk) || attn_small_shape(d, Nq, Nk));
}

extern "C" int edadm_attention_fused_f16(const void* Q, int64_t ldq, int64_t strideQ, int64_t headQ, const void* K, int64_t ldk,
                                         int64_t strideK, int64_t headK, const void* V, int64_t ldv, int64_t strideV,
                                         int64_t headV, void* out, int64_t ldo, int64_t strideO,
                                         int64_t B, int64_t heads, int64_t Nq, int64_t Nk, int64_t d, float alpha_qk,
                                         const float* pqp, float alpha_pv, int out_mode, const float* oqp, void* stream) {
    if (!Q || !K || !V || !out || !pqp || B <= 0 || !edadm_attention_fused_ok(heads, d, Nq, Nk)) return EDADM_EINVAL;
    if ((out_mode != 0 && out_mode != 2) || (out_mode == 2 && !oqp)) return EDADM_EINVAL;
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (strideQ & 7) || (strideK & 7) || (strideV & 7) || (headQ & 7) || (headK & 7) ||
        (headV & 7) || headQ < d || headK < d || headV < d)
        return EDADM_EINVAL;
    if (((uintptr_t)Q & 15) || ((uintptr_t)K & 15) || ((uintptr_t)V & 15)) return EDADM_EINVAL;
    if (out_mode == 0 && (((uintptr_t)out & 15) || (ldo & 3) || (strideO & 3) || (d & 3))) return EDADM_EINVAL;
    if (out_mode == 2 && (((uintptr_t)out & 3) || (ldo & 3) || (strideO & 3))) return EDADM_EINVAL;
    if (!(alpha_qk > 0.f)) return EDADM_EINVAL;
    const QP* pq = reinterpret_cast<const QP*>(pqp);
    hipStream_t st = (hipStream_t)stream;
    if (attn_small_shape(d, Nq, Nk)) {
        const dim3 grid((unsigned)((Nq + 63) / 64), (unsigned)heads, (unsigned)B);
#define ATTS_LAUNCH(KD32_, NT_, DH_)                                                                                               \
        hipLaunchKernelGGL((k_attn_small<KD32_, NT_, DH_>), grid, dim3(256), 0, st, (const __half*)Q, ldq, strideQ, headQ, (const __half*)K, \
                           ldk, strideK, headK, (const __half*)V, ldv, strideV, headV, out, ldo, strideO, (int)Nq, (int)Nk, alpha_qk, pq,   \
                           alpha_pv, out_mode, reinterpret_cast<const QP*>(oqp))
        if (d == 576) ATTS_LAUNCH(18, 16, 1);
        else ATTS_LAUNCH(30, 4, 2);
#undef ATTS_LAUNCH
        return edadm_launch_status();
    }
    if (attn_wide_shape(d, Nq, Nk)) {
        static const int64_t form = EDADM_TUNE_I("EDADM_ATTN_WIDE_FORM", 16);
        if (form == 16) {
            hipLaunchKernelGGL((k_attn_wide16<12, 24>), dim3((unsigned)((Nq + ATT_BQ - 1) / ATT_BQ), (unsigned)heads, (unsigned)B), dim3(512), 0, st,
                               (const __half*)Q, ldq, strideQ, headQ, (const __half*)K, ldk, strideK, headK, (const __half*)V, ldv, strideV,
                               headV, out, ldo, strideO, (int)Nq, (int)Nk, alpha_qk, pq, alpha_pv, out_mode, reinterpret_cast<const QP*>(oqp));
            return edadm_launch_status();
        }
        hipLaunchKernelGGL((k_attn_wide<24, 12>), dim3((unsigned)((Nq + ATT_BQ - 1) / ATT_BQ), (unsigned)heads, (unsigned)B), dim3(256), 0, st,
                           (const __half*)Q, ldq, strideQ, headQ, (const __half*)K, ldk, strideK, headK, (const __half*)V, ldv, strideV,
                           headV, out, ldo, strideO, (int)Nq, (int)Nk, alpha_qk, pq, alpha_pv, out_mode, reinterpret_cast<const QP*>(oqp));
        return edadm_launch_status();
    }
    const int kd = (int)((d + 15) / 16), dvb = (int)((d + 31) / 32);
#define ATT_CASE(KD_, DVB_)                                                                                                  \
    if (kd == KD_ && dvb == DVB_) {                                                                                          \
        hipLaunchKernelGGL((k_attn_fused<KD_, DVB_>), dim3((unsigned)((Nq + ATT_BQ - 1) / ATT_BQ), (unsigned)heads, (unsigned)B), \
                           dim3(256), 0, st, (const __half*)Q, ldq, strideQ, headQ, (const __half*)K, ldk, strideK, headK,             \
                           (const __half*)V, ldv, strideV, headV, out, ldo, strideO, (int)Nq, (int)Nk, (int)d, alpha_qk, pq, alpha_pv, out_mode,                  \
                           reinterpret_cast<const QP*>(oqp));                                                                        \
        return edadm_launch_status();                                                                                        \
    }
    ATT_CASE(1, 1) ATT_CASE(2, 1) ATT_CASE(3, 2) ATT_CASE(4, 2) ATT_CASE(5, 3) ATT_CASE(6, 3) ATT_CASE(7, 4) ATT_CASE(8, 4)
    ATT_CASE(9, 5) ATT_CASE(10, 5)
#undef ATT_CASE
    return EDADM_EINVAL;
}
