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
#include "../../include/edadm.h"
#include <hip/hip_fp16.h>

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
                worst = fmaxf(worst, fabsf(t - r_[r]));
            }
            if (__builtin_expect(worst > 0.499f, 0)) {
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

extern "C" int edadm_attention_fused_ok(int64_t heads, int64_t d, int64_t Nq, int64_t Nk) {
    return heads >= 1 && d >= 8 && d <= 160 && (d & 7) == 0 && Nq >= 1 && Nk >= 2;
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
