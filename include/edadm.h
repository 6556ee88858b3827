/* edadm.h — C ABI of the MI355X-native EDA-DM hot path (libedadm.so, gfx950 only).
 *
 * The reference (BienLuky/EDA-DM) has no FFI layer: its hot path is chains of stock PyTorch ops
 * inside qdiff/ *.py and the UNet definitions.  Each entry point below replaces one such chain
 * (SURVEY.md §2 "K#" table) and cites the reference file:line it stands for.  The Python host
 * (eda-dm_amd/qdiff, .../edadm) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer borrowed from the caller (kept alive until the stream
 *    op completes); no torch types; sizes are element counts unless said otherwise;
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it
 *    (graph-capturable: no allocation, no synchronisation inside);
 *  - return 0 on success, negative errno-style code otherwise (-22 bad argument,
 *    -5 launch failure); nothing throws;
 *  - float arithmetic that decides integer codes is IEEE fp32 with true division and
 *    round-half-to-even, exactly like the reference (`round(x / delta)`).
 */
#ifndef EDADM_H
#define EDADM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int edadm_abi_version(void);

/* ---- K1: uniform affine fake-quant -------------------------------------------------------
 * qdiff/quant_layer.py:266-276 (UniformAffineQuantizer.forward, inited path).
 * out = (clamp(rint(x/delta)+zp, 0, qmax) - zp) * delta ; per-tensor (nq==1) or per-row-block
 * (nq>1: parameter index = (i / inner) % nq, i.e. dim-0 channel-wise for weights).
 * Training "prob" mix (`where(rand < prob, xq, x)`, :271-272): if u!=NULL the uniforms are
 * taken from u (injected mask, parity tests); else if prob<1 a counter RNG keyed by
 * (seed, element index) is used; prob>=1 disables mixing.  codes (optional, may be NULL)
 * receives the unsigned integer code as float. */
int edadm_fake_quant_fwd(const float* x, float* out, float* codes, int64_t n,
                         const float* delta, const float* zp, int64_t nq, int64_t inner,
                         float qmax, const float* u, float prob, uint64_t seed, void* stream);

/* Backward of the above for per-tensor delta (LSQ-style): quant_layer.py:19-23,266-276 through
 * autograd.  gx = (gy*delta*inrange)/delta [masked-out elements pass gy]; gdelta (1 float) =
 * sum gy*(code-zp) - (gy*delta*inrange)*((x/delta)/delta).  ws: >= edadm_reduce_ws_floats()
 * floats of scratch. */
int edadm_fake_quant_bwd(const float* gy, const float* x, float* gx, float* gdelta, int64_t n,
                         const float* delta, const float* zp, float qmax,
                         const float* u, float prob, uint64_t seed, float* ws, void* stream);
int64_t edadm_reduce_ws_floats(void);

/* ---- K2: AdaRound ---------------------------------------------------------------------------
 * qdiff/adaptive_rounding.py:49-72.  Weight viewed as rows x cols with leading dimensions
 * (split weights are column slices of the full tensor, quant_layer.py:424-427); alpha/galpha are
 * dense rows*cols; delta/zp per row.  soft!=0: floor(w/d)+clamp(sigmoid(a)*1.2-0.1,0,1);
 * soft==0: floor(w/d)+(a>=0). */
int edadm_adaround_init_alpha(const float* w, int64_t ldw, float* alpha, int64_t rows, int64_t cols,
                              const float* delta, void* stream);
int edadm_adaround_fwd(const float* w, int64_t ldw, const float* alpha, float* out, int64_t ldo,
                       int64_t rows, int64_t cols, const float* delta, const float* zp, float qmax,
                       int soft, void* stream);
int edadm_adaround_bwd(const float* gy, int64_t ldg, const float* w, int64_t ldw, const float* alpha,
                       float* galpha, int64_t rows, int64_t cols, const float* delta, const float* zp,
                       float qmax, void* stream);

/* ---- K3: MSE clip-range search scores ---------------------------------------------------------
 * qdiff/quant_layer.py:120-147,150-213: score[c] = mean |q_c(x) - x|^2.4 for candidate c with
 * (scale[c], zp[c]); per-tensor: x[n], nc candidates -> score[nc];
 * per-channel: x[rows][cols], candidates laid out [nc][rows] -> score[nc][rows]. nc <= 128. */
int edadm_mse_scores_tensor(const float* x, int64_t n, const float* scale, const float* zp, int nc,
                            float qmax, float* score, float* ws, void* stream);
int edadm_mse_scores_channel(const float* x, int64_t rows, int64_t cols, const float* scale,
                             const float* zp, int nc, float qmax, float* score, void* stream);
int edadm_minmax(const float* x, int64_t n, float* out2, float* ws, void* stream);
/* K3 bookkeeping with IEEE arithmetic in the reference's operation order: candidate (scale, zp) grids
 * from per-row (min, max) — mode 1: 1-D search, num thresholds (one_side -1/0/+1 = neg/no/pos);
 * mode 2: 2-D search, num x 2^bits (channel_clamp: the per-channel variant clamps the range to
 * include 0, quant_layer.py:125-126) — laid out [nc][rows]; then the first-minimum selection, the
 * 0.1/0.9 EMA of the range for activations (run_min/run_max state, first != 0 on the first batch,
 * quant_layer.py:79-85) and the final delta / zero_point (quant_layer.py:95-105). */
int edadm_mse_candidates(const float* xmin, const float* xmax, int64_t rows, int mode, int one_side,
                         int n_bits, int num, int channel_clamp, float* scale, float* zp, void* stream);
int edadm_mse_select(const float* score, int64_t nc, int64_t rows, const float* xmin, const float* xmax,
                     int mode, int one_side, int n_bits, int num, int channel_clamp, float* run_min,
                     float* run_max, int first, float* delta, float* zp, void* stream);

/* ---- K7: reconstruction loss -------------------------------------------------------------------
 * qdiff/quant_layer.py:26-33 with p=2, reduction 'none': sum((pred-tgt)^2) / (numel / C).
 * fwd writes 1 float; bwd: gpred = gscale[0] * 2 (pred-tgt) / (numel/C). */
int edadm_lp_loss_fwd(const float* pred, const float* tgt, int64_t n, float inv_denom, float* loss,
                      float* ws, void* stream);
int edadm_lp_loss_bwd(const float* pred, const float* tgt, int64_t n, float inv_denom,
                      const float* gscale, float* gpred, void* stream);
/* The per-module terms of the fine-grained loss (qdiff_control/block_recon.py:186-189, qdiff/block_recon.py:176-180: add_loss *
 * lp_loss(module_q[j], module_r[j], p = 2)) as a gradient injection -- only their gradient is ever used.  ONE pass over the gradient
 * arriving at a hooked module's output [rows_total][row_elems] (memory order):
 *   gin[r] = gout[r] + (row0 <= r < row0 + nrows ? 2 inv_denom gscale[0] (pred[r] - tgt[idx ? idx[r - row0] : r - row0]) : 0)
 * tgt: the FP feature rows (the cached per-sample maps, gathered through the minibatch indices idx[nrows], or the rows themselves
 * when idx is null); row_elems % 4 == 0, 16-byte aligned pointers.  Replaces gather + edadm_lp_loss_bwd + autograd's zero-padded
 * slice gradient + accumulation add with the same two fp32 operations per element in the same order (same bits). */
int edadm_lp_loss_inject(const float* gout, const float* pred, const float* tgt, const int64_t* idx, int64_t rows_total,
                         int64_t row0, int64_t nrows, int64_t row_elems, float inv_denom, const float* gscale, float* gin,
                         void* stream);

/* ---- K8: fused Adam step (torch.optim.Adam semantics as used at block_recon.py:112-117,199-206)
 * hyper = device float[4]: {lr/bias_corr1, sqrt(bias_corr2), beta1, beta2}; eps fixed 1e-8. */
int edadm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                    void* stream);

/* ---- K10: stochastic input mixing  block_recon.py:141-145 ------------------------------------ */
int edadm_mix_where(const float* a, const float* b, float* out, int64_t n, const float* u, float prob,
                    uint64_t seed, void* stream);

/* ---- K9: DDIM update -------------------------------------------------------------------------
 * ddim/functions/denoising.py:50-56 and ldm/models/diffusion/ddim_control.py:221-253.
 * Per-sample coefficient table coef[B][5] = {sqrt(1-a_t), sqrt(a_t), sqrt(a_prev),
 * sqrt(1-a_prev-sigma^2), sigma}; classifier-free guidance:
 * e = e_u + s (e_c - e_u) when e_uncond != NULL. x, e: [B][chw]. */
int edadm_ddim_step(const float* x, const float* e_cond, const float* e_uncond, float cfg_scale,
                    const float* coef, const float* noise, float* x_prev, float* pred_x0,
                    int64_t B, int64_t chw, void* stream);
/* K9b: PLMS update (ldm/models/diffusion/plms.py:205-279): CFG combine -> e_t (stored when e_t != NULL, it is
 * the multistep history), e' by `order` (0: e_t; -1: (old1 + e_t)/2, second half of the first pseudo improved
 * Euler step; 1..3: Adams-Bashforth over old1 (newest) .. old3), then pred_x0 / x_prev with coef as above. */
int edadm_plms_step(const float* x, const float* e_cond, const float* e_uncond, float cfg_scale,
                    const float* old1, const float* old2, const float* old3, int order, const float* coef,
                    float* e_t, float* x_prev, float* pred_x0, int64_t B, int64_t chw, void* stream);

/* ---- activation quantisation to MFMA operands (inference form of K1) --------------------------
 * code = clamp(rint(x/delta)+zp,0,qmax); i8 operand = code-128; f16 operand = code-zp.
 * qp = device float[4*nseg]: {delta, zp, qmax, unused} per channel segment; channels
 * [0,split) use segment 0, [split,C) segment 1 (quant_layer.py:415-419). x is [rows][C]. */
int edadm_quant_i8(const float* x, int8_t* out, int64_t rows, int64_t C, const float* qp, int64_t split,
                   void* stream);
/* the same over the channel concatenation [x1 | x2] (C1 and C2 channels; x2 may be NULL): the UNet skip
 * concatenation (openaimodel.py:778, diffusion.py:320) is consumed in place, never materialised in fp32 */
int edadm_quant_i8_cat(const float* x1, int64_t C1, const float* x2, int64_t C2, int8_t* out, int64_t rows,
                       const float* qp, int64_t split, void* stream);
int edadm_quant_f16(const float* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int64_t C,
                    const float* qp, float premul, void* stream);
/* NCHW fp32 -> NHWC (boundary transposes of QuantModel.forward, quant_model.py:69). */
int edadm_nchw_to_nhwc(const float* x, float* out, int64_t B, int64_t C, int64_t HW, void* stream);
int edadm_nhwc_to_nchw(const float* x, float* out, int64_t B, int64_t C, int64_t HW, void* stream);
/* im2col for tiny-Cin convolutions (conv_in, Cin=3/4): NHWC fp32 -> quantised [M][Kpad] i8 */
int edadm_im2col_quant_i8(const float* x, int8_t* out, int64_t B, int64_t H, int64_t W, int64_t C,
                          int64_t Kpad, const float* qp, void* stream);

/* ---- K5: GroupNorm (+SiLU) (+quantise) on NHWC -------------------------------------------------
 * ddim/models/diffusion.py:27-33,121-128; ldm/modules/diffusionmodules/util.py:199-216;
 * openaimodel.py:201-228.  stats: mean/rstd per (b, group) into stats[B][G][2].
 * apply: y = (x-mean)*rstd*gamma+beta [* (1+scale[b][c]) + shift[b][c]] [-> silu] then either
 * fp32 out and/or up to 3 quantised i8 outputs with their own (delta,zp,qmax) (q,k,v convs share
 * one normalised input, quant_block.py:419-424). */
int64_t edadm_gn_ws_floats(int64_t B, int64_t HW, int64_t C);
int edadm_groupnorm_stats(const float* x, float* stats, float* ws, int64_t B, int64_t HW, int64_t C, int64_t G,
                          float eps, void* stream);
int edadm_groupnorm_apply(const float* x, const float* stats, const float* gamma, const float* beta,
                          const float* scale_shift, int64_t B, int64_t HW, int64_t C, int64_t G, int silu,
                          float* out_f32, int8_t* q0, int8_t* q1, int8_t* q2, const float* qp, int nq,
                          void* stream);
/* both passes over the channel concatenation [x1 | x2] (x2 may be NULL), C = C1 + C2 */
int edadm_groupnorm_stats_cat(const float* x1, int64_t C1, const float* x2, int64_t C2, float* stats, float* ws,
                              int64_t B, int64_t HW, int64_t G, float eps, void* stream);
int edadm_groupnorm_apply_cat(const float* x1, int64_t C1, const float* x2, int64_t C2, const float* stats,
                              const float* gamma, const float* beta, const float* scale_shift, int64_t B, int64_t HW,
                              int64_t G, int silu, float* out_f32, int8_t* q0, int8_t* q1, int8_t* q2,
                              const float* qp, int nq, void* stream);
/* the same with one more output: qraw = the UN-normalised input quantised with qp_raw (two quantisers over the channel
 * ranges [0, raw_split) / [raw_split, C) when raw_split > 0) = what edadm_quant_i8_cat writes for the second consumer of
 * the tensor -- the ResBlock's skip convolution (openaimodel.py:265-277) -- without reading it again */
int edadm_groupnorm_apply_cat_raw(const float* x1, int64_t C1, const float* x2, int64_t C2, const float* stats,
                                  const float* gamma, const float* beta, const float* scale_shift, int64_t B, int64_t HW,
                                  int64_t G, int silu, float* out_f32, int8_t* q0, int8_t* q1, int8_t* q2,
                                  const float* qp, int nq, int8_t* qraw, const float* qp_raw, int64_t raw_split,
                                  int64_t B2, void* stream);
/* B2 / rows2 > 0 (the *_rep forms and edadm_groupnorm_apply_cat_raw): x2 holds B2 < B images (rows2 < rows rows) and is read
 * periodically -- image b of the concatenation takes x2's image b % B2.  Serves the skip tensors a classifier-free-guidance
 * pair shares: both halves of the doubled batch are identical up to the first context-dependent layer, so the sampling
 * loop evaluates that prefix once per pair and its skip tensors stay at half the batch. */
int edadm_groupnorm_stats_cat_rep(const float* x1, int64_t C1, const float* x2, int64_t C2, float* stats, float* ws,
                                  int64_t B, int64_t HW, int64_t G, float eps, int64_t B2, void* stream);
int edadm_quant_i8_cat_rep(const float* x1, int64_t C1, const float* x2, int64_t C2, int8_t* out, int64_t rows,
                           const float* qp, int64_t split, int64_t rows2, void* stream);
/* pass 2 alone: per-channel partials [B][nchunk][C][2] (sum, sum of squares) written by a producer's epilogue
 * (edadm_qconv3_i8_direct) -> stats; ws2 (may be NULL) holds the second half of a channel concatenation */
int edadm_groupnorm_final_cat(const float* ws1, int64_t C1, const float* ws2, int64_t C2, float* stats, int64_t B,
                              int64_t HW, int64_t G, int64_t nchunk, float eps, void* stream);
/* the same with the second partials buffer holding B2 < B images, read periodically (image b takes b % B2): the shared
 * half of a classifier-free-guidance pair; B2 = 0: same batch */
int edadm_groupnorm_final_cat_rep(const float* ws1, int64_t C1, const float* ws2, int64_t C2, float* stats, int64_t B,
                                  int64_t HW, int64_t G, int64_t nchunk, float eps, int64_t B2, void* stream);
/* the same with a slab count of its own for the second buffer (nchunk2 = 0: as the first): producers that cut an image into
 * 64-row and 32-row slabs (edadm_qconv3_i8_direct with 256- and 128-pixel tiles) on the two halves of a concatenation */
int edadm_groupnorm_final_cat_rep2(const float* ws1, int64_t C1, const float* ws2, int64_t C2, float* stats, int64_t B,
                                   int64_t HW, int64_t G, int64_t nchunk1, int64_t nchunk2, float eps, int64_t B2, void* stream);
/* LayerNorm over the last dim, same output options (ldm/modules/attention.py:222-242). */
int edadm_layernorm_quant(const float* x, const float* gamma, const float* beta, int64_t rows, int64_t C,
                          float eps, float* out_f32, int8_t* q0, int8_t* q1, int8_t* q2, const float* qp,
                          int nq, void* stream);
/* LayerNorm of x[row] + radd[row / rows_per_batch] (C % 4 == 0); the sum is written to sum_out [rows][C] -- the updated
 * residual stream: edadm_add_rowbcast folded into the norm that consumes its result.  xrows: rows of x (a divisor of
 * `rows`: x is read periodically -- the shared half of a guidance pair -- or rows itself) */
int edadm_layernorm_quant_radd(const float* x, int64_t xrows, const float* radd, int64_t rows_per_batch, float* sum_out,
                               const float* gamma, const float* beta, int64_t rows, int64_t C, float eps, int8_t* q0,
                               int8_t* q1, int8_t* q2, const float* qp, int nq, void* stream);
/* SiLU / GEGLU producers (quant_block.py:86-116 emb path; attention.py:37-45). */
int edadm_silu_quant_i8(const float* x, int8_t* out, int64_t n, const float* qp, void* stream);
int edadm_geglu_quant_i8(const float* x, int8_t* out, int64_t rows, int64_t inner, const float* qp,
                         void* stream);
int edadm_silu(const float* x, float* out, int64_t n, void* stream);
int edadm_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* out[m][c] = x[m][c] + r[m / rows_per_batch][c] (C % 4 == 0): the cross-attention branch over a ONE-token context
 * (class-conditional LDM, ldm/modules/attention.py:168-194 with context [B,1,D]) is one vector per image -- softmax over
 * a single key is exactly 1 for every query -- so it is computed for one query row per image and broadcast */
int edadm_add_rowbcast(const float* x, const float* r, float* out, int64_t rows, int64_t C, int64_t rows_per_batch,
                       void* stream);
/* the same with x of xrows < rows rows read periodically (out has `rows` rows): the shared half of a guidance pair */
int edadm_add_rowbcast_rep(const float* x, const float* r, float* out, int64_t rows, int64_t C, int64_t rows_per_batch,
                           int64_t xrows, void* stream);
int edadm_concat_c(const float* a, int64_t Ca, const float* b, int64_t Cb, float* out, int64_t rows,
                   void* stream);
int edadm_avgpool2_nhwc(const float* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
int edadm_upsample2_nhwc(const float* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);

/* ---- K4: quantised conv / linear on int8 MFMA ---------------------------------------------------
 * qdiff/quant_layer.py:406-437 at inference: out[m][n] = scale[n]*sum_k a[m][k]*w[n][k] + bias[n]
 * (+ rowadd[m / rows_per_batch][n], rows_per_batch >= 16) (+ residual[m][n]) with a = code-128 (int8), w = wcode-zp_w
 * (int8); the (128-zp_x)*sum_k w term and delta_x*delta_w[n] are folded into bias/scale by the
 * host.  geom (device-independent ints, 16 of them):
 *  {mode, B, H, W, Cin, Ho, Wo, KH, KW, stride, pad0, upsample, padval, 0,0,0}
 *  mode 0: dense rows A[m][lda]; mode 1: implicit-GEMM convolution over NHWC int8 input,
 *  K ordered [ky][kx][ci]; padval = int8 operand of real zero (zp_x-128). */
int edadm_qgemm_i8(const int8_t* A, int64_t lda, const int8_t* Wt, int64_t ldw, int64_t M, int64_t N,
                   int64_t K, const int32_t* geom, const float* scale, const float* bias,
                   const float* rowadd, int64_t rows_per_batch, const float* residual, int64_t ldr,
                   float* out, int64_t ldo, void* stream);
/* edadm_qgemm_i8 that also writes the NEXT GroupNorm's partial sums (ldm/modules/diffusionmodules/util.py GroupNorm32 over the
 * tensor this layer produces: openaimodel.py:217-263 ResBlock.in_layers[0] after a SpatialTransformer's proj_out
 * (attention.py:247-275) or a Downsample convolution (openaimodel.py:143-170)): gn_ws [M / 64][N][2] = per-channel (sum, sum of
 * squares) of every 64-row slab of out (fp32 of the stored values, one fixed order of sums -- the direct convolution's, so
 * edadm_groupnorm_final_cat* reduces both alike); hw = rows per image.  Full tiles only: edadm_qgemm_i8_gn_ok (M % 256 == 0,
 * N % 192 == 0 or N % 128 == 0, hw % 64 == 0, M % hw == 0). */
int edadm_qgemm_i8_gn_ok(int64_t M, int64_t N, int64_t hw);
int edadm_qgemm_i8_gn(const int8_t* A, int64_t lda, const int8_t* Wt, int64_t ldw, int64_t M, int64_t N,
                      int64_t K, const int32_t* geom, const float* scale, const float* bias,
                      const float* rowadd, int64_t rows_per_batch, const float* residual, int64_t ldr,
                      float* out, int64_t ldo, float* gn_ws, int64_t hw, void* stream);
/* Deferred device-side errors.  Every entry point returns as soon as its kernels are enqueued, so a failure INSIDE a kernel
 * (today: a hand-off wait of the persistent GEMM that gave up after 4 M polls instead of hanging the GPU -- its outputs are
 * then wrong) cannot come back from the launching call: it sets a device error word.  This call synchronises `stream`, reads
 * the word and returns 0 or -EIO; clear != 0 resets it.  Callers check it where they synchronise anyway (end of a sampling
 * run, end of a calibration unit).  Never call it inside a stream capture. */
int edadm_device_status(int clear, void* stream);
/* Per-device set-up of the library's constant tables (the padding-row table the LDS-DMA gathers of the GEMM / convolution
 * kernels read "padding" from), filled synchronously for the CURRENT device.  Call once per device before the first stream
 * capture of the process: the lazy fill on a first launch is a synchronous copy, which is illegal while another stream is in a
 * global-mode capture.  The reference has no counterpart (its padding is F.conv2d's, quant_layer.py:434).  Idempotent. */
int edadm_init_device(void);
/* Diagnostics for the measurement tools (tools/unet_prof.py -> tools/pmc_traffic.py): the kernel structures the int8 GEMM /
 * convolution entry points launched on this host thread since the last call, in launch order, as tags (1 k_gemm_nt, 2 k_gemm_nt8,
 * 3 k_gemm_p, 4 k_gemm_ntq, 5 k_conv3_direct, 6 k_gemm_split2, 7 k_gemm_br); writes up to 8 of them, returns their number and forgets them.
 * Lets a tool attribute each layer's ALGORITHMIC bytes to the kernel that ran it, so the PMC traffic per kernel name has its own
 * denominator.  No reference counterpart. */
int edadm_diag_launch_kernels(int32_t* tags8);
/* Batched f16 NT GEMM for the attention products (integer-valued f16 operands, exact in fp32):
 * C[z][m][n] = alpha * sum_k A[z][m][k] * B[z][n][k], z = outer*inner + head with two-level
 * element strides (outer = batch sample, inner = attention head inside a [B][N][heads*d] tensor).
 * quant_block.py:427-446,204-235; openaimodel.py:384-406. */
int edadm_gemm_f16_nt(const void* A, int64_t lda, int64_t strideA, int64_t strideA_i, const void* Bm,
                      int64_t ldb, int64_t strideB, int64_t strideB_i, float* C, int64_t ldc,
                      int64_t strideC, int64_t strideC_i, int64_t batch, int64_t inner, int64_t M,
                      int64_t N, int64_t K, float alpha, void* stream);
/* edadm_qgemm_i8's contract with f16 operands (a = code - zp_x, w = wcode - zp_w, exact integers):
 * for layers whose integer weights do not fit int8 (8-bit weights whose zero point is 127). */
int edadm_qgemm_f16(const void* A, int64_t lda, const void* Wt, int64_t ldw, int64_t M, int64_t N,
                    int64_t K, const int32_t* geom, const float* scale, const float* bias,
                    const float* rowadd, int64_t rows_per_batch, const float* residual, int64_t ldr,
                    float* out, int64_t ldo, void* stream);
/* A dense layer with SPLIT quantisers in ONE launch -- the 1x1 skip convolution of an up-path ResBlock over the concatenation
 * [h | skip] (quant_layer.py:415-427: act_quantizer / act_quantizer_0 and weight_quantizer / weight_quantizer_0 over the channel
 * ranges [0, split) and [split, K1 + K2), one F.conv2d over the concatenated operands; split set by quant_block.py:86-116 /
 * openaimodel.py:746-783).  A: int8 operand [M][lda], columns [0, K1) quantised by the first activation quantiser, [split = K1, K1 + K2)
 * by the second; W1 [N][ldw1], W2 [N][ldw2]: integer weights of the two ranges; scale1 / scale2 [N] = delta_x * delta_w per range;
 * bias [N] (zero-point corrections folded in) or NULL.  out fp32 [M][ldo] = fl(fl(scale2 acc2) + fl(scale1 acc1 + bias)): the bits
 * of two edadm_qgemm_i8 launches with the second accumulating through the residual port, without writing the fp32 output
 * three times.  Shapes: edadm_qgemm_i8_split2_ok (M % 128 == 0, N % 192 == 0, K1 % 64 == K2 % 64 == 0). */
int edadm_qgemm_i8_split2_ok(int64_t M, int64_t N, int64_t K1, int64_t K2);
int edadm_qgemm_i8_split2(const int8_t* A, int64_t lda, int64_t split, const int8_t* W1, int64_t ldw1, const int8_t* W2, int64_t ldw2,
                          int64_t M, int64_t N, int64_t K1, int64_t K2, const float* scale1, const float* scale2, const float* bias,
                          float* out, int64_t ldo, void* stream);
/* Variants whose epilogue feeds the consuming activation quantizer directly (no fp32 round trip through
 * HBM): out_mode 1 -> f16 operand (code - zp) [M][N]; 2 -> int8 operand (code - 128) [M][N]; 3 (i8 only)
 * -> GEGLU a*gelu(gate) over INTERLEAVED (a_j, gate_j) output columns, then int8 operand [M][N/2]
 * (ldm/modules/attention.py:37-45 followed by quant_layer.py:266-269); 4 (i8 only) -> the f16 operand of mode 1
 * stored TRANSPOSED per image, out[b][n][m - b*rows_per_batch] with row length ldo: the [d][Nk] B operand of the
 * attention P.V product written by the v projection itself (rows_per_batch % 32 == 0, M % 128 == 0, N a multiple
 * of the 128/192 tile, no rowadd / residual).  oqp = device float[3] {delta, zero_point, qmax} of that quantizer;
 * ldo counts output elements. */
int edadm_qgemm_i8_q(const int8_t* A, int64_t lda, const int8_t* Wt, int64_t ldw, int64_t M, int64_t N,
                     int64_t K, const int32_t* geom, const float* scale, const float* bias,
                     const float* rowadd, int64_t rows_per_batch, const float* residual, int64_t ldr,
                     void* out, int64_t ldo, int out_mode, const float* oqp, void* stream);
/* GROUPED form of edadm_qgemm_i8_q: `count` (1..4) dense layers of the same M and K in ONE launch, each with its own int8 operand A
 * [M][lda], integer weights W [N][ldw], per-column scale / bias, output, output mode (1..4 as above) and consuming quantiser oqp -- the
 * q / k / v projections of a self-attention (ldm/modules/attention.py:168-176; three QuantModules, quant_layer.py:406-437, whose input
 * quantisers may differ after reconstruction: three operands, not one), or one GEGLU projection (attention.py:37-45; count = 1).
 * Kernels k_gemm_br / k_gemm_bw (csrc/gemm.hip): a workgroup keeps a 192-column weight block resident in LDS and streams the
 * activation rows; several MFMA waves per SIMD work on different row tiles so that one wave's quantising epilogue overlaps another's
 * matrix work (br: two groups of four waves + loader waves; bw: twelve independent waves that fetch their own rows).  The
 * codes are those of `count` separate edadm_qgemm_i8_q launches, bit for bit.  Shapes: edadm_qgemm_i8_grouped_q_ok (M % 128 == 0,
 * every N % 192 == 0, K = 384 or 576, enough tiles to fill the chip); no rowadd, no residual.  `probs` is a HOST array (its fields are
 * copied into the launch). */
typedef struct edadm_gemm_problem {
    const int8_t* A;
    int64_t lda;
    const int8_t* W;
    int64_t ldw;
    const float* scale;
    const float* bias;            /* may be NULL */
    void* out;
    int64_t ldo;                  /* output elements per row (mode 3: N / 2 bytes; mode 4: row length of the transposed image) */
    const float* oqp;             /* device float[3] {delta, zero_point, qmax} */
    int64_t N;
    int64_t rows_per_batch;       /* mode 4 only */
    int32_t out_mode;
    int32_t reserved;
} edadm_gemm_problem;
int edadm_qgemm_i8_grouped_q_ok(int64_t M, int64_t N, int64_t K);
int edadm_qgemm_i8_grouped_q(const edadm_gemm_problem* probs, int count, int64_t M, int64_t K, void* stream);
/* The same attention (quant_block.py:204-235; openaimodel.py:384-406) for ONE WIDE head with q and k as INT8 operands (code - 128:
 * edadm_qgemm_i8_q out_mode 2) and v as the f16 operand (out_mode 1): the score product runs on the int8 MFMA, exact in int32;
 * zq = zero point of the q quantiser (the (128 - zq) sum_d k8 term is added per key; the key-independent terms cancel in the softmax).
 * Strides in ELEMENTS of each tensor's own type.  Shapes: edadm_attention_fused_i8qk_ok (d == 384, Nk % 64 == 0).  Output as
 * edadm_attention_fused_f16 (out_mode 0: fp32, 2: the consumer's int8 operand). */
int edadm_attention_fused_i8qk_ok(int64_t heads, int64_t d, int64_t Nq, int64_t Nk);
int edadm_attention_fused_i8qk(const int8_t* Q, int64_t ldq, int64_t strideQ, int64_t headQ, const int8_t* K, int64_t ldk, int64_t strideK,
                               int64_t headK, const void* V, int64_t ldv, int64_t strideV, int64_t headV, void* out, int64_t ldo,
                               int64_t strideO, int64_t B, int64_t heads, int64_t Nq, int64_t Nk, int64_t d, float alpha_qk, float zq,
                               const float* pqp, float alpha_pv, int out_mode, const float* oqp, void* stream);
/* K6f: the whole quantised attention core in one kernel -- S = alpha_qk Qc Kc^T, P = softmax(S), Pc = quantise(P; pqp), O = alpha_pv
 * Pc Vc -- for heads of 8 <= d <= 160 (d % 8 == 0) and any number of queries / keys >= 2: the heads x Nq x Nk score matrix is
 * never written (quant_block.py:204-235, :119-162, :398-451; openaimodel.py:384-406).  Q [B][Nq][..], K / V [B][Nk][..]: f16 integer
 * codes minus zero point, head h at columns h*head*.. of its tensor (d for [.., heads*d]; 3d for the legacy (q|k|v)-per-head
 * layout, openaimodel.py:390-393; ld*, batch and head strides in elements, multiples of 8);
 * out [B][Nq][heads*d]: fp32 (out_mode 0) or the consumer's int8 operand (out_mode 2, oqp).  edadm_attention_fused_ok: shape gate. */
int edadm_attention_fused_ok(int64_t heads, int64_t d, int64_t Nq, int64_t Nk);
/* the legacy AttentionBlock's qkv tensor [rows][heads x (q|k|v) x d] quantised to f16 codes in one pass, same layout: column c
 * takes quantiser qp3[(c / d) % 3] and that group's pre-multiplier (q * scale, k * scale: openaimodel.py:391-399) */
int edadm_quant_f16_qkv(const float* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int64_t C, int64_t d,
                        const float* qp3, float premul_q, float premul_k, float premul_v, void* stream);
int edadm_attention_fused_f16(const void* Q, int64_t ldq, int64_t strideQ, int64_t headQ, const void* K, int64_t ldk,
                              int64_t strideK, int64_t headK, const void* V, int64_t ldv, int64_t strideV, int64_t headV,
                              void* out, int64_t ldo, int64_t strideO,
                              int64_t B, int64_t heads, int64_t Nq, int64_t Nk, int64_t d, float alpha_qk,
                              const float* pqp, float alpha_pv, int out_mode, const float* oqp, void* stream);
int edadm_gemm_f16_nt_q(const void* A, int64_t lda, int64_t strideA, int64_t strideA_i, const void* Bm,
                        int64_t ldb, int64_t strideB, int64_t strideB_i, void* C, int64_t ldc,
                        int64_t strideC, int64_t strideC_i, int64_t batch, int64_t inner, int64_t M,
                        int64_t N, int64_t K, float alpha, int out_mode, const float* oqp, void* stream);
/* ---- fp32 contraction of the calibration graph (H1) -----------------------------------------------
 * quant_layer.py:434 `F.conv2d / F.linear` on fake-quantised operands and its autograd backward
 * (block_recon.py:197): C[z] = alpha * A[z] . B[z]^T (+ bias[n]) (+ residual[m][n]) on the exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32); convolutions through im2col (forward), col2im (input gradient) and a
 * split-K transposed product + slab sum (weight gradient); all NHWC, deterministic. */
/* fp32 operand -> two-term f16 expansion for the three-product fp32 emulation on the f16 MFMA:
 * x [R][T][C] fp32 (C % 4 == 0) -> out [R][T][3][C] f16 = (hi, lo, hi) per K group for order 0 (operand A) or
 * (hi, hi, lo) for order 1 (operand B), of x * 2^e with |x * 2^e| < 2^14: one exponent per row (per_row, operand B:
 * a row = one output channel) or one for the tensor.  inv[r] (inv[0]) = 2^-e.  comb (per-tensor mode only):
 * comb[n] = inv[0] * other[n] (other[0] when n_other == 1) = the per-column factor edadm_qgemm_f16 applies to
 * the accumulators.  ws = edadm_reduce_ws_floats() floats.  The contraction itself is edadm_qgemm_f16 over K = 3 T C.
 * amax_parts: NULL, or the 1024 partial maxima edadm_absmax_parts wrote for this tensor (an operand that is
 * expanded more than once -- forward and weight gradient -- is scanned once).
 * order 2 (C % 16 == 0): out [R][T][C / 16][2][16] = every 16 k-values as [hi x16 | lo x16] (2 T C f16 per row) -- the
 * operand layout of edadm_qgemm_f16x3 / edadm_gemm_f16x3_nt, whose kernels form the three products from the two
 * fragments, so that each term crosses memory, L2 and LDS once. */
int edadm_absmax_parts(const float* x, int64_t n, float* parts, void* stream);
int edadm_split_f16(const float* x, int64_t R, int64_t T, int64_t C, int order, int per_row, const float* amax_parts,
                    void* out, float* inv, const float* other, int64_t n_other, float* comb, int64_t N, float* ws,
                    void* stream);
/* GroupNorm (+ swish) of x [B][HW][C] fp32 NHWC written DIRECTLY as the order-2 expansion out [B HW][C / 16][2][16] f16 (what
 * edadm_groupnorm_stats + _apply + edadm_absmax_parts + edadm_split_f16 produce in four passes over the normalised tensor, in
 * two over x): the expansion's power-of-two scale comes from a bound on max|y| computed from per-channel minima / maxima of x
 * gathered next to the moment partials (y is monotone in x per channel) -- the same exponent as the scan of y unless the two
 * straddle a power of two.  The first-stage decoder's norm -> nonlinearity -> conv chain (reference:
 * ldm/modules/diffusionmodules/model.py:124-136 ResnetBlock.forward norm1/norm2 -> swish -> conv, :566-568 Decoder.forward norm_out).  inv / other / comb as
 * edadm_split_f16.  C % 16 == 0, C <= 1024, B * G <= 1024; ws = edadm_gn_split_ws_floats(B, HW, C, G) floats, 16-byte aligned. */
int64_t edadm_gn_split_ws_floats(int64_t B, int64_t HW, int64_t C, int64_t G);
int edadm_gn_split_f16(const float* x, int64_t B, int64_t HW, int64_t C, int64_t G, float eps, const float* gamma,
                       const float* beta, int silu, void* out, float* inv, const float* other, int64_t n_other, float* comb,
                       int64_t N, float* ws, void* stream);
/* the weight gradient's operands in one pass: in [R][C] fp32 -> out [C][R / L][3][L] f16 (transposed, the reduction
 * axis R cut into slabs of L rows, each slab the (hi, lo, hi) / (hi, hi, lo) expansion of order 0 / 1 under one
 * power-of-two scale for the tensor; inv[0] = 1 / scale).  R % L == 0, L even, R C % 4 == 0.  The product is
 * edadm_gemm_f16_nt over batch = R / L slabs of K = 3 L, summed by edadm_sum_slabs.
 * geom = host {B, H, W, C, Ho, Wo, KH, KW, stride, pad} (C % 64 == 0; amax_parts required): `in` is the NHWC activation
 * and the matrix is its im2col, gathered on the fly (R = B Ho Wo, C = KH KW C) -- the convolution's weight gradient
 * without the [M][KH KW C] matrix.
 * order 2 (L % 16 == 0): out [C][R / L][L / 16][2][16], for edadm_gemm_f16x3_nt over slabs of K2 = 2 L.
 * ldo = f16 elements between output rows (>= 2 R or 3 R, % 8 == 0): callers pad it off powers of two, which would put
 * every output row on the same HBM channels. */
int edadm_transpose_split_f16(const float* in, int64_t R, int64_t C, int64_t L, int order, const int32_t* geom,
                              const float* amax_parts, void* out, int64_t ldo, float* inv, float* ws, void* stream);
/* The contraction over order-2 expansions (operand type 3 of the K4 template: per 64 operand bytes one hi.hi, one lo.hi
 * and one hi.lo f16 MFMA, fp32 accumulation).  edadm_qgemm_f16x3: edadm_qgemm_f16's contract (out = acc * scale[n] +
 * bias[n] (+ residual)), K2 = 2 K f16 per row (K2 % 32 == 0), geom[4] = 2 C f16 per pixel (C % 16 == 0).
 * edadm_gemm_f16x3_nt: C[z] = alpha * A[z] . B[z]^T over `batch` slabs (element strides), for the weight gradient. */
int edadm_qgemm_f16x3(const void* A, int64_t lda, const void* Wt, int64_t ldw, int64_t M, int64_t N, int64_t K2,
                      const int32_t* geom, const float* scale, const float* bias, const float* residual, int64_t ldr,
                      float* out, int64_t ldo, void* stream);
int edadm_gemm_f16x3_nt(const void* A, int64_t lda, int64_t strideA, const void* Bm, int64_t ldb, int64_t strideB,
                        float* C, int64_t ldc, int64_t strideC, int64_t batch, int64_t M, int64_t N, int64_t K2,
                        float alpha, void* stream);
int edadm_gemm_f32_nt(const float* A, int64_t lda, int64_t strideA, const float* Bm, int64_t ldb,
                      int64_t strideB, float* C, int64_t ldc, int64_t strideC, int64_t batch, int64_t M,
                      int64_t N, int64_t K, float alpha, const float* bias, const float* residual,
                      int64_t ldr, void* stream);
int edadm_im2col_f32(const float* x, float* cols, int64_t B, int64_t H, int64_t W, int64_t C, int64_t Ho,
                     int64_t Wo, int KH, int KW, int stride, int pad, void* stream);
int edadm_col2im_f32(const float* dcols, float* dx, int64_t B, int64_t H, int64_t W, int64_t C, int64_t Ho,
                     int64_t Wo, int KH, int KW, int stride, int pad, void* stream);
int edadm_sum_slabs(const float* slabs, float* out, int64_t n, int64_t S, void* stream);
/* fp32 3x3 / pad-1 convolution with few output channels (the network's last layer, whose
 * activation quantizer is disabled, quant_model.py:90-95): x NHWC fp32, w [N][3][3][C] fp32. */
int edadm_conv3x3_f32_smalln(const float* x, const float* w, const float* bias, float* out, int64_t B,
                             int64_t H, int64_t W, int64_t C, int64_t N, void* stream);
/* Codebook lookup of the VQ first stage (VQModelInterface.decode -> self.quantize, ldm/models/autoencoder.py:274-277; taming's
 * VectorQuantizer2 is not vendored by the reference: its published rule, parity unpinned): out[r] = codebook[argmin_j |z_r - e_j|^2],
 * first minimum, d = |z|^2 + |e|^2 - 2 z.e in fp32.  z, out [R][D], codebook [E][D], D <= 8; idx (optional) [R] int64. */
int edadm_vq_nearest(const float* z, const float* codebook, float* out, int64_t* idx, int64_t R, int64_t D, int64_t E,
                     void* stream);
/* K6: row softmax + quantise to f16 codes (code - zp); rows x cols fp32 in. */
int edadm_softmax_quant_f16(const float* s, void* out, int64_t rows, int64_t cols, int64_t ldo,
                            const float* qp, void* stream);
/* f16 [b][n][d] (ld) -> [b][d][npad] transpose for the PV product's B operand */
int edadm_transpose_f16(const void* x, int64_t ldx, int64_t strideX, void* out, int64_t ldo,
                        int64_t strideO, int64_t batch, int64_t n, int64_t d, void* stream);
/* weight packing: int4 nibble codes <-> int8 operand (code - zp_row) */
/* ---- fp32 NHWC graph pieces of the first-stage decoder (ldm/modules/diffusionmodules/model.py:35-205,465-572) ----
 * convolution as an implicit GEMM on the exact-fp32 MFMA: x [B][H][W][C] (C % 4 == 0), w [N][KH][KW][C], zero padding,
 * optional nearest-2x upsample of x folded into the gather (Upsample, model.py:43-60), out [B][Ho][Wo][N]
 * (+bias[n]) (+residual[m][n]); plain row softmax of the attention block (model.py:193-195). */
int edadm_conv2d_f32_nhwc(const float* x, const float* w, const float* bias, const float* residual, float* out,
                          int64_t B, int64_t H, int64_t W, int64_t C, int64_t Ho, int64_t Wo, int64_t N, int KH, int KW,
                          int stride, int pad, int ups, void* stream);
int edadm_softmax_f32(const float* s, float* out, int64_t rows, int64_t cols, void* stream);
/* K4w: the int8 x int4 GEMM on PACKED weights (two 4-bit codes per byte as edadm_pack_w4 writes them, row n at byte n*K/2, code -
 * zp4[n] = the int8 operand): nibbles expanded in registers into the int8 MFMA, zero points taken off through the row sums of A --
 * the bits of edadm_qgemm_i8 on the unpacked weights at half the weight bytes.  For few-row layers (M <= ~2048: time-embedding
 * tables, one-token context branches) where the weights are the traffic.  K % 32 == 0.  quant_layer.py:406-437. */
int edadm_qgemm_w4(const int8_t* A, int64_t lda, const uint8_t* W4, const float* zp4, int64_t M, int64_t N, int64_t K,
                   const float* scale, const float* bias, const float* rowadd, int64_t rows_per_batch,
                   const float* residual, int64_t ldr, float* out, int64_t ldo, void* stream);
int edadm_unpack_w4(const uint8_t* packed, const float* zp, int8_t* out, int64_t rows, int64_t cols,
                    void* stream);
int edadm_pack_w4(const int8_t* w, const float* zp, uint8_t* packed, int64_t rows, int64_t cols, void* stream);

/* Mask-RNG epoch (a device word added, times an odd constant, to the seed argument of edadm_fake_quant_fwd/_bwd and
 * edadm_mix_where): add = 0 sets it, add = 1 increments it by `value`.  A reconstruction iteration replayed from a HIP graph
 * (block_recon.py:136-210 as one graph launch) keeps its captured seed arguments and bumps the epoch at the head of every
 * replay, so each iteration draws fresh masks (the `torch.rand_like` of quant_layer.py:271-272, block_recon.py:141-143). */
int edadm_rng_epoch(uint64_t value, int add, void* stream);

/* Direct 3x3 convolution (stride 1, pad 1) of the long-K int8 layers (quant_layer.py:406-437 at inference, F.conv2d): the
 * input patch of a 256-pixel tile stays in LDS per 64-channel chunk, so an activation byte crosses L2 -> LDS once per
 * chunk instead of once per tap.  Wdc = the filter in the kernel's own layout (edadm_conv3_pack_w from the [N][3][3][Cin]
 * int8 filter of edadm_qgemm_i8); same epilogue contract as edadm_qgemm_i8 (scale, bias, per-image row-add, fp32 residual).
 * edadm_conv3_direct_tile: the output pixels per workgroup tile for (B, H, W, Cin, N) -- 256, or 128 where 256-pixel tiles
 * would not fill one round of the CUs (the 8x8 level) or do not divide the batch -- and 0 when the shape is not one the kernel takes
 * (W in {8,16,32,64}, H a power of two, Cin % 64 == 0, N % 192 == 0, whole tiles); edadm_conv3_direct_ok: tile != 0.
 * H, W are the dimensions the convolution runs over; with ups = 1 the stored
 * tensor is [B][H/2][W/2][Cin] and its nearest-2x upsample (openaimodel.py:110-118, diffusion.py:41-45) is read in place.
 * gn_ws (or NULL): [M / 64][N][2] per-channel (sum, sum of squares) of every 64-row slab of the output, written from the
 * epilogue's registers in an order that does not depend on the tile -- the partials edadm_groupnorm_final_cat reduces
 * (H * W % 64 == 0). */
/* rows of the packed filter edadm_conv3_pack_w writes for N output channels (N rounded up to the kernel's 192- or 128-column
 * block; the padding rows are zero filters): the caller allocates rows * 9 * Cin bytes.  0: no direct kernel for this N. */
int64_t edadm_conv3_packed_rows(int64_t N);
int edadm_conv3_pack_w(const int8_t* w, int8_t* out, int64_t N, int64_t Cin, void* stream);
int edadm_conv3_direct_ok(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t N);
int edadm_conv3_direct_tile(int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t N);
int edadm_qconv3_i8_direct(const int8_t* A, const int8_t* Wdc, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t N,
                           int padval, int ups, const float* scale, const float* bias, const float* rowadd, int64_t rows_per_batch,
                           const float* residual, int64_t ldr, float* out, int64_t ldo, float* gn_ws, void* stream);
/* The same kernel on the two-term f16 expansions of fp32 operands (edadm_split_f16 order 2): the 3x3 / stride 1 /
 * pad 1 convolutions of the calibration graph (quant_layer.py:434 on fake-quantised fp32 operands, forward and input gradient) and
 * of the first-stage decoder (model.py:465-572) with the input patch resident in LDS -- an activation crosses L2 -> LDS once per
 * 16-channel chunk instead of nine times.  A [B][H][W][C / 16][hi x16 | lo x16] f16 (with ups = 1: stored at [H/2][W/2]), Wdc = the filter
 * [N][3][3][C / 16][hi x16 | lo x16] f16 packed by edadm_conv3_pack_w(w, out, N, 4 C) (bytes: rows of 4 C per tap);
 * out[m][n] = comb[n] * (a_hi b_hi + a_hi b_lo + a_lo b_hi) + bias[n] (+ residual).  C % 16 == 0; shapes: edadm_conv3_f16x3_direct_ok --
 * W in {8 .. 64} as the int8 form, and 128 .. 1024 (the decoder's upper levels: images cut into 64-column blocks). */
int edadm_conv3_f16x3_direct_ok(int64_t B, int64_t H, int64_t W, int64_t C, int64_t N);
int edadm_qconv3_f16x3_direct(const void* A, const void* Wdc, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int ups,
                              const float* comb, const float* bias, const float* residual, int64_t ldr, float* out, int64_t ldo,
                              void* stream);

/* ---- H1 training-graph ops: forward and input-gradient of the non-contraction ops of the calibration graph, fp32, on the
 * reference's layouts (csrc/train_ops.hip).  GroupNorm (+ SiLU) over NCHW, `stats` = [B * G][2] (mean, rstd) written by the
 * forward and read by the backward (ddim/models/diffusion.py:27-35, openaimodel.py:215-223, quant_block.py:86-116,321-348);
 * LayerNorm over [rows][C], C <= 4096 (attention.py:201-203); GEGLU over [rows][2 inner] (attention.py:37-45); SiLU and
 * softmax gradients (quant_block.py:204-235); a batched [Z][R][C] -> [Z][C][R] transpose for the K-major operands of the
 * attention products' gradients.  The normalisation affines receive no gradient: the loop never trains them
 * (block_recon.py:44-108). */
/* The same GroupNorm (+ SiLU) forward / input gradient over NHWC x [B][HW][C] (C % 4 == 0, C <= 1024), the layout of the
 * contraction kernels: a convolutional unit's reconstruction iteration then runs without NCHW <-> NHWC passes between its
 * operators.  ws = edadm_gn_nhwc_ws_floats(B, HW, C, G) floats (per-chunk per-channel partial sums, reduced in fp64). */
int64_t edadm_gn_nhwc_ws_floats(int64_t B, int64_t HW, int64_t C, int G);
int edadm_gn_fwd_nhwc(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* ws, int64_t B,
                      int64_t C, int64_t HW, int G, float eps, int silu, void* stream);
int edadm_gn_bwd_nhwc(const float* dy, const float* x, const float* gamma, const float* beta, const float* stats, float* dx,
                      float* ws, int64_t B, int64_t C, int64_t HW, int G, int silu, void* stream);
int edadm_gn_fwd_nchw(const float* x, const float* gamma, const float* beta, float* y, float* stats, int64_t B, int64_t C,
                      int64_t HW, int G, float eps, int silu, void* stream);
int edadm_gn_bwd_nchw(const float* dy, const float* x, const float* gamma, const float* beta, const float* stats, float* dx,
                      int64_t B, int64_t C, int64_t HW, int G, int silu, void* stream);
int edadm_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int64_t rows, int64_t C,
                 float eps, void* stream);
int edadm_ln_bwd(const float* dy, const float* x, const float* gamma, const float* stats, float* dx, int64_t rows, int64_t C,
                 void* stream);
int edadm_geglu_fwd(const float* h, float* out, int64_t rows, int64_t inner, void* stream);
int edadm_geglu_bwd(const float* dy, const float* h, float* dh, int64_t rows, int64_t inner, void* stream);
int edadm_silu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int edadm_softmax_bwd(const float* dp, const float* p, float* dx, int64_t rows, int64_t cols, void* stream);
int edadm_softmax_fwd_any(const float* s, float* out, int64_t rows, int64_t cols, void* stream);
int edadm_transpose_batched_f32(const float* x, float* out, int64_t Z, int64_t R, int64_t C, void* stream);

/* ---- TDAC step scores (scripts/calibration.py:47-69: the O(T^2) density / variety loop over the mid-block features) -------------
 * feats [T][B][C][P] fp32 (feature map of sampling step t, NCHW with P = H W).  For every pair: mse[i][j] = mean((F_i - F_j)^2)
 * (the reference tests it against the density radius, :49-52) and cosdis[i][j] = sum over (b, p) of 1 - cos(F_i[b, :, p], F_j[b, :, p])
 * with each norm clamped at eps (nn.CosineSimilarity(dim=1, eps=1e-6), :57-63).  Both [T][T], symmetric, zero diagonal; one launch,
 * deterministic.  The host then counts / adds over j in the reference's order (edadm/tdac.py). */
int edadm_tdac_pair_scores(const float* feats, int64_t T, int64_t B, int64_t C, int64_t P, float eps, float* mse, float* cosdis,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif
