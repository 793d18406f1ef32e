/*
 * drakes_hip.h — C ABI of libdrakes_hip.so: hand-written gfx950 (MI355X) kernels for the
 * masked-diffusion reverse-sampling hot path of DRAKES (DNA).
 *
 * The reference has no FFI: its boundary is the Python method surface of `Diffusion`
 * (drakes_dna/diffusion_gosai_update.py).  Each entry point below names the reference code it
 * replaces; drakes_amd/diffusion.py re-assembles them behind the reference's method names and
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing here allocates, frees or
 *     synchronises; `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return 0 on success, a negative DRK_E* code otherwise (drk_last_error() has the text);
 *   - activations are channels-last fp32 [B][L][128]; tokens are int64 [B][L] with ids
 *     A,C,G,T = 0..3 and mask = 4 (dataloader_gosai.py:11, diffusion_gosai_update.py:82-88);
 *   - safe to call concurrently on distinct streams (no global mutable state except the
 *     thread-local error string).
 *
 * Noise convention (the build's own; the reference draws torch.rand_like at :27/:33): when the
 * `uniforms` argument of a step function is NULL the kernel evaluates Philox4x32-10 with
 *   counter = (position, sample & 0xffffffff, step, 0x44524B53 ^ (sample >> 32)), key = seed,
 *   U[v] = (r[v] >> 8) * 2^-24 (v<4),  U[4] = ((r0&255)<<16 | (r1&255)<<8 | (r2&255)) * 2^-24,
 * where sample = sample_offset + row.  Rows therefore draw the same noise however the batch
 * is sharded over GPUs.
 */
#ifndef DRAKES_HIP_H
#define DRAKES_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRK_OK 0
#define DRK_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define DRK_EHIP (-2)     /* a HIP runtime call or launch failed */

int drk_version(void);
const char* drk_last_error(void);

/* ---- noise -------------------------------------------------------------------------------- */
/* Writes the step's uniforms, float32 [B][L][5].  Test/injection helper: what the step kernels
 * evaluate in-register when `uniforms == NULL`.  Stands in for torch.rand_like at
 * diffusion_gosai_update.py:27,33. */
int drk_step_uniforms(float* u, int B, int L, uint64_t seed, uint32_t step, uint64_t sample_offset,
                      void* stream);

/* ---- reverse step (hard) ------------------------------------------------------------------ */
/* SUBS log-probs from raw backbone logits: Diffusion._subs_parameterization,
 * diffusion_gosai_update.py:243-253.  logits,logp float32 [B][L][5] (may alias); x int64 [B][L],
 * or NULL with x_soft float32 [B][L][5] (held token = argmax, :247-249). */
int drk_subs_fwd(const float* logits, const int64_t* x, const float* x_soft, float* logp, int B, int L,
                 void* stream);

/* One hard reverse step from raw logits: _subs_parameterization + the tail of _ddpm_update
 * (:478-486) + _sample_categorical (:24-28).  x_next may alias x.  mc_t/mc_s are the step's
 * move chances when all batch rows sit at the same time (as _sample builds t, :527-528);
 * mc_rows (optional, device float32 [B][2] = (mc_t, mc_s) per row) overrides them row by row:
 * _ddpm_update accepts any t [B,1] (:462-474). */
int drk_reverse_step_hard(const float* logits, const int64_t* x, const float* uniforms, int64_t* x_next,
                          int B, int L, float mc_t, float mc_s, const float* mc_rows, uint64_t seed,
                          uint32_t step, uint64_t sample_offset, void* stream);

/* Noise-removal: SUBS, then argmax over the 4 nucleotides — _sample, :541-549. */
int drk_denoise_argmax(const float* logits, const int64_t* x, int64_t* x_out, int B, int L, void* stream);

/* The same step in a form a captured launch sequence (hipGraph) can replay for every step: (mc_t, mc_s) of step k are
 * mc_table[2k], mc_table[2k+1] (device, float32 [num_steps][2]) and the step index is *step_counter (device int),
 * which this call advances by one after the step (a separate one-thread launch); rng = device uint64 [2] =
 * (seed, sample_offset).  In-kernel Philox noise only.
 * Same arithmetic as drk_reverse_step_hard (diffusion_gosai_update.py:462-486 + the loop of :524-539). */
int drk_reverse_step_hard_table(const float* logits, const int64_t* x, int64_t* x_next, int B, int L,
                                const float* mc_table, int* step_counter, const uint64_t* rng, void* stream);

/* ---- wide vocabulary: the protein reverse step ------------------------------------- */
/* One step of Interpolant.sample, drakes_protein/fmif/fm_utils.py:113-134, from the model's logits float32 [B][L][V]
 * (V = 22: 20 amino acids, X, mask = index 21): temperature, mask logit at -1e6, log-softmax, held tokens kept,
 * q = exp(logp)*d_t with q[mask] = move_chance_s, Gumbel-argmax (_sample_categorical, :14-18), carry-over.
 * x, x_next int64 [B][L] (may alias); pred_clean (optional) int64 [B][L] = argmax of the non-mask logits (:115-117).
 * uniforms: float32 [B][L][V] or NULL = Philox blocks (position, sample, step, 0x50524F54 ^ sample_hi ^ block<<24),
 * class c = word c%4 of block c/4. */
int drk_reverse_step_hard_wide(const float* logits, const int64_t* x, const float* uniforms, int64_t* x_next,
                               int64_t* pred_clean, int B, int L, int V, int mask_index, float temp, float d_t,
                               float move_chance_s, uint64_t seed, uint32_t step, uint64_t sample_offset, void* stream);

/* The Gumbel-softmax step of Interpolant.sample_gradient, drakes_protein/fmif/fm_utils.py:159-192 (soft branch) with
 * _sample_categorical_gradient (:21-26): x_soft, x_next float32 [B][L][V]; flag float32 [B][L] = 1 - x_soft[mask].
 * q gets the reference's +1e-8 floor (:181).  V = 22. */
int drk_reverse_step_soft_wide_fwd(const float* logits, const float* x_soft, const float* uniforms, float* x_next,
                                   float* flag, int B, int L, int V, int mask_index, float temp, float d_t,
                                   float move_chance_s, float gumbel_temp, uint64_t seed, uint32_t step,
                                   uint64_t sample_offset, void* stream);
/* Its backward: given grad_x_next [B][L][V] (and optionally grad_flag [B][L]), writes dlogits and the direct dx; the
 * forward row is recomputed from (logits, x_soft, noise). */
int drk_reverse_step_soft_wide_bwd(const float* logits, const float* x_soft, const float* uniforms,
                                   const float* grad_x_next, const float* grad_flag, float* dlogits, float* dx, int B,
                                   int L, int V, int mask_index, float temp, float d_t, float move_chance_s,
                                   float gumbel_temp, uint64_t seed, uint32_t step, uint64_t sample_offset, void* stream);

/* ---- likelihood -------------------------------------------------------------- */
/* q(x_t | x_0): Diffusion.q_xt, diffusion_gosai_update.py:425-436.  x0, xt int64 [B][L]; a position is masked when its
 * uniform < move_chance.  uniforms: float32 [B][L] (ONE per position, as torch.rand(*x.shape) there) or NULL = the
 * first Philox uniform of (seed, step, sample_offset + row, position).  mc_rows (optional, device float32 [B]): one
 * move chance per row (the reference takes move_chance [B,1]) instead of the scalar. */
int drk_q_xt(const int64_t* x0, const float* uniforms, int64_t* xt, int B, int L, float move_chance,
             const float* mc_rows, uint64_t seed, uint32_t step, uint64_t sample_offset, void* stream);
/* One time step of Diffusion.get_likelihood, diffusion_gosai_update.py:911-917: acc[b][l] += multiplier *
 * log p(x0[b][l] | x_t) from RAW backbone logits [B][L][5] (SUBS applied inside).  acc float32 [B][L]. */
int drk_likelihood_accum(const float* logits, const int64_t* x0, const int64_t* xt, float multiplier, float* acc,
                         int B, int L, void* stream);

/* ---- denoiser (CNNModel.forward, models/dnaconv.py:74-110), fp32 ---------------------------- */
/* relu(Linear(GaussianFourierProjection(sigma))) — dnaconv.py:19-21,50,82.
 * sigma [nb]; W [64]; lin_w [128][128]; lin_b [128]; temb [nb][128]. */
int drk_time_embed_fwd(const float* sigma, int nb, const float* W, const float* lin_w, const float* lin_b,
                       float* temb, void* stream);
/* The n_layers per-layer Dense projections of the time embedding — dnaconv.py:59,92.
 * dense_w [n_layers][128][128]; dense_b [n_layers][128]; tbias [nb][n_layers][128]. */
int drk_time_bias_fwd(const float* temb, int nb, const float* dense_w, const float* dense_b, int n_layers,
                      float* tbias, void* stream);
/* Stem: relu(Conv1d(5->128,k=9,pad=4)(one_hot(x))) — dnaconv.py:76-77,83-84.
 * x int64 [B][L] or (x == NULL) x_soft float32 [B][L][5]; w_t [9][5][128] (= weight.permute(2,1,0));
 * feat [B][L][128]. */
int drk_stem_fwd(const int64_t* x, const float* x_soft, const float* w_t, const float* bias, float* feat,
                 int B, int L, void* stream);
/* h = LayerNorm_width(feat + tbias[:, None, :]) — dnaconv.py:90-95.  tbias row stride
 * `tbias_bstride` floats between batch rows (0 = one row shared by the whole batch).
 * mean/rstd [B*L] are optional outputs for the backward pass.
 * width = args.hidden_dim (dnaconv.py:60), a multiple of 4 up to 128.  Rows are always 128 floats: a narrower model
 * lives zero-padded in the first `width` lanes (gamma = beta = 0 beyond them), the statistics run over `width`. */
int drk_ln_fwd(const float* feat, const float* tbias, long tbias_bstride, const float* gamma,
               const float* beta, float* h, float* mean, float* rstd, int B, int L, int width, void* stream);
/* feat_out = relu(Conv1d(128->128,k=taps,dilation)(h) + bias) [+ resid] — dnaconv.py:96-98 and,
 * with taps=1 and resid=NULL, final_conv.0 + ReLU (:61-62,101).  Exact fp32 (f32-input MFMA).
 * w_pk [taps][32][128][4]: w_pk[t][g][co][q] = weight[co][4g+q][t].  bias may be NULL (= 0).
 * relu_mask: optional uint8 [B][L][128] = (pre-activation > 0), kept for the backward pass.
 * The same kernel is the convolution's input gradient: feed dc, the weights packed from
 * weight.transpose(0,1).flip(2), bias = NULL, relu = 0. */
int drk_conv_fwd_f32(const float* h, const float* w_pk, const float* bias, const float* resid, float* out,
                     unsigned char* relu_mask, int B, int L, int taps, int dilation, int relu, void* stream);
/* The same operation as drk_conv_fwd_f32 (dnaconv.py:96-98; taps = 1: final_conv.0, :61-62,101) at fp32 GRADE on the
 * bf16 matrix cores: weights and activations are each taken as three bf16 planes (x = x0 + x1 + x2 covers fp32's 24
 * bits) and a product as the six partial products of plane sum <= 2, accumulated in fp32 — the result differs from the
 * exact-fp32 kernel's by summation order and by dropped terms of <= 2^-24 per product.  h stays fp32 (split in
 * registers); w3 = the split weights, bf16 [taps][3 planes][4 k-steps][4][2][64][8] in the fragment order of
 * drk_layer_bf16's stream (drakes_amd/ops.py::pack_conv_weight_x3).  h and out must be distinct; out may alias resid.
 * Also the input gradient, as for drk_conv_fwd_f32. */
int drk_conv_fwd_x3(const float* h, const void* w3, const float* bias, const float* resid, float* out,
                    unsigned char* relu_mask, int B, int L, int taps, int dilation, int relu, void* stream);
/* logits = Conv1d(128->5,k=1)(y) — dnaconv.py:63,101-102.  w [5][128]; logits [B][L][5]. */
int drk_head_out_fwd(const float* y, const float* w, const float* bias, float* logits, int B, int L,
                     void* stream);

/* ---- backward ops (fp32) -----------------------------------------------------------------------
 * The reference gets these from torch autograd over CNNModel.forward (models/dnaconv.py:74-110),
 * _subs_parameterization (:243-253) and _ddpm_update_finetune_gradient (:552-586).  Parameter
 * gradients are ACCUMULATED into buffers the caller zeroes (the conv weight gradient by an ordered reduction,
 * the small vectors by fp32 atomics). */
/* out = g * mask (ReLU backward from the mask of drk_conv_fwd_f32); n % 4 == 0. */
int drk_mask_mul(const float* g, const unsigned char* mask, float* out, long n, void* stream);
/* out = g * (y > 0) (ReLU backward from a saved ReLU output); n % 4 == 0. */
int drk_relu_bwd(const float* g, const float* y, float* out, long n, void* stream);
/* out[seg][128] += column sums of x[seg*seg_rows .. +seg_rows][128] (bias / time-bias gradients). */
int drk_colsum128(const float* x, float* out, long n_seg, long seg_rows, void* stream);
/* LayerNorm backward of drk_ln_fwd: dxin (optional output), dfeat = dxin + g_resid (g_resid may be NULL), dgamma/dbeta
 * accumulated; dtbias_shared (optional, float32 [128], accumulated): the column sum of dxin = the gradient of a time bias
 * that one row serves the whole batch with (tbias_bstride == 0).  Optional fused hand-over to the layer below (the bf16
 * path): next_mask = that layer's ReLU mask (uint8 [B*L][128]), dc_next = bf16(dfeat * next_mask) (bf16 [B*L][128]),
 * dbias_next (float32 [128], accumulated) = its column sums, the conv bias gradient. */
int drk_ln_bwd(const float* dh, const float* feat, const float* tbias, long tbias_bstride, const float* gamma,
               const float* mean, const float* rstd, const float* g_resid, float* dxin, float* dfeat,
               float* dgamma, float* dbeta, float* dtbias_shared, const unsigned char* next_mask, void* dc_next,
               float* dbias_next, int B, int L, int width /* as in drk_ln_fwd */, void* stream);
/* Conv1d weight gradient: dwt[t][ci][co] += sum_rows h[row + (t - taps/2)*dilation][ci] * dc[row][co].  dbias (optional, float32 [128], accumulated): column sums of dc,
 * the conv bias gradient, taken by the centre-tap workgroups on the way.  Deterministic split-K: every (row chunk, tap)
 * workgroup stores its 128x128 partial into `workspace` (device memory, 16-byte aligned, at least
 * drk_conv_wgrad_workspace_bytes(B, L, taps) bytes, contents irrelevant before and after) and a second launch adds the
 * partials in chunk order: the same bits on every run.  Calls sharing one workspace must be ordered (one stream). */
long drk_conv_wgrad_workspace_bytes(int B, int L, int taps);   /* host-side size query; launches nothing */
int drk_conv_wgrad_f32(const float* h, const float* dc, float* dwt, float* dbias, int B, int L, int taps, int dilation,
                       void* workspace, long workspace_bytes, void* stream);
/* drk_conv_wgrad_f32 without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE: it
 * exists for callers that cannot allocate or carry a workspace — code under hipGraph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  The partial slabs live in a 33.3 MB array inside the library's code object (one
 * copy per GPU, held until the process ends).  Same arithmetic, same chunking, hence the same bits as
 * drk_conv_wgrad_f32.  Limits: taps <= 9, B*L <= 4 194 304 rows (DRK_EINVAL beyond).  Every call on a GPU shares that
 * GPU's array: calls that overlap on one GPU (different streams) race; calls ordered on one stream are safe. */
int drk_conv_wgrad_f32_static(const float* h, const float* dc, float* dwt, float* dbias, int B, int L, int taps,
                              int dilation, void* stream);
/* Backward of drk_head_out_fwd: dy [B][L][128]; dw [5][128], db [5] accumulated. */
int drk_head_out_bwd(const float* dlogits, const float* y, const float* w, float* dy, float* dw, float* db,
                     int B, int L, void* stream);
/* Backward of drk_stem_fwd from dpre = dfeat * (feat > 0): dw_t [9][5][128] accumulated; dx [B][L][5]
 * (NULL for token input: no gradient flows into integer tokens). */
int drk_stem_bwd(const float* dpre, const int64_t* x, const float* x_soft, const float* w_t, float* dw_t,
                 float* dx, int B, int L, void* stream);
/* The stem's weight and bias gradients (models/dnaconv.py:80-86 under torch autograd) of n_seg recorded backward calls of
 * one shape in ONE launch: dw_t [9][5][128] +=, db [128] += (NULL: not wanted).  dpre_ptrs, x_ptrs, x_soft_ptrs: DEVICE
 * arrays of n_seg pointers — each call's dpre [B*L][128] and its input, tokens int64 [B*L] (x_ptrs) or soft float32
 * [B*L][5] (x_soft_ptrs): exactly one of the two arrays, the other NULL. */
int drk_stem_wgrad_multi(const float* const* dpre_ptrs, const int64_t* const* x_ptrs, const float* const* x_soft_ptrs,
                         int n_seg, float* dw_t, float* db, int B, int L, void* stream);
/* Backward of drk_time_bias_fwd + drk_time_embed_fwd from dtbias [nb][n_layers][128]; all outputs accumulated.
 * dtemb: float32 [nb][128] scratch, ZEROED by the caller (the layers' contributions to the embedding gradient meet there). */
int drk_time_bwd(const float* dtbias, const float* sigma, int nb, const float* W, const float* temb,
                 const float* dense_w, int n_layers, float* d_dense_w, float* d_dense_b, float* d_lin_w,
                 float* d_lin_b, float* dtemb, void* stream);
/* Backward of drk_subs_fwd: rows holding a token are constants (dlogits = 0), :250-252. */
int drk_subs_bwd(const float* logits, const int64_t* x, const float* x_soft, const float* dlogp, float* dlogits,
                 int B, int L, void* stream);

/* ---- reverse step (soft, Gumbel-softmax) --------------------------------------------------------
 * _subs_parameterization + the tail of _ddpm_update_finetune_gradient (:572-586) +
 * _sample_categorical_gradient (:30-35).  x_soft, x_next float32 [B][L][5]; flag [B][L] is the soft copy
 * flag (:577-581); copy_flag_temp <= 0 means None.  The backward recomputes the forward row from the
 * same inputs and returns dL/dlogits and the DIRECT part of dL/dx_soft (the part through the
 * denoiser input comes from drk_stem_bwd).  mc_rows: optional per-row (mc_t, mc_s), as in
 * drk_reverse_step_hard. */
int drk_reverse_step_soft_fwd(const float* logits, const float* x_soft, const float* uniforms, float* x_next,
                              float* flag, int B, int L, float mc_t, float mc_s, const float* mc_rows, float temp,
                              float copy_flag_temp, uint64_t seed, uint32_t step, uint64_t sample_offset,
                              void* stream);
int drk_reverse_step_soft_bwd(const float* logits, const float* x_soft, const float* uniforms,
                              const float* grad_x_next, const float* grad_flag, float* dlogits, float* dx, int B,
                              int L, float mc_t, float mc_s, const float* mc_rows, float temp, float copy_flag_temp,
                              uint64_t seed, uint32_t step, uint64_t sample_offset, void* stream);

/* ---- fine-tune loss: per-step KL term (caller of the path) ------------------------
 * kl[b] = sum_{l, v<4} flag[b,l] * (-p + p_old + p*(lp - lp_old)) / move_chance_t * last_x[b,l,v], p = exp(lp):
 * finetune_reward_bp.py:67-74 (`kl_div` of one reverse step; the caller sums the steps, :83).  logp, logp_old,
 * last_x float32 [B][L][5] (column 4 unused); flag = copy_flag [B][L], float32 (soft steps) or int64 (hard steps),
 * exactly one of the two pointers non-NULL; kl float32 [B]. */
int drk_kl_step_fwd(const float* logp, const float* logp_old, const float* last_x, const float* flag_f,
                    const int64_t* flag_i, float move_chance_t, float* kl, int B, int L, void* stream);
/* Gradients of sum_b grad_kl[b] * kl[b]: dlogp [B][L][5] (column 4 = 0) always; dlogp_old [B][L][5], dlast_x [B][L][5]
 * and dflag [B][L] when non-NULL (soft steps, where last_x and copy_flag carry a graph,
 * diffusion_gosai_update.py:580-586, and through last_x so does old_model.forward(last_x), finetune_reward_bp.py:68). */
int drk_kl_step_bwd(const float* logp, const float* logp_old, const float* last_x, const float* flag_f,
                    const int64_t* flag_i, const float* grad_kl, float move_chance_t, float* dlogp, float* dlogp_old,
                    float* dlast_x, float* dflag, int B, int L, void* stream);

/* ---- fused sampling loop, bf16 MFMA ----------------------------------------------------------- */
/* The whole of Diffusion._sample (diffusion_gosai_update.py:506-550, ddpm predictor) in one launch:
 * num_steps x (CNNModel.forward + _subs_parameterization + _ddpm_update tail) and, if
 * noise_removal, the final forward + argmax (:541-549).  One workgroup per sequence; L <= 224.
 *   wstream   bf16 weight stages: n_layers*9 conv taps + final_conv.0 (layout: drakes_amd/fastpath.py::pack_stream)
 *   stem_w    bf16 [3][4][2][64][8] stem weights as MFMA fragments over K = 8*tap + v (fastpath.py::_frag_stem), head2_w bf16 [8][128] final_conv.2 weights (same file)
 *   vecs      float32 [n_layers+1][3][128] biases and LayerNorm affine terms (same file)
 *   tbias     float32 [n_tb][n_layers][128] time biases (drk_time_bias_fwd); forward f reads row
 *             f * tb_fwd_stride / (n_layers*128) (stride 0: time_conditioning=False)
 *   mc_table  float32 [num_steps][2] = (move_chance_t, move_chance_s) per step
 *   uniforms  NULL (in-kernel Philox) or float32 [num_steps][B][L][5]
 *   x_init    NULL (all-mask prior, :438-440) or int64 [B][L];  x_out int64 [B][L]
 *   traj_out  NULL or int8 [num_steps (+1 with noise removal)][B][L]: the state every forward starts from (what
 *             _sample_finetune_gradient records as last_x / copy_flag on its hard steps, :615-619)
 *   logits_out NULL or float32 [B][L][5]: raw logits of forward number dump_fwd (tests)
 *   dbg_feat  NULL or float32 [B][L][128]: residual stream of forward dump_fwd after layer
 *             dbg_layer (-1 = after the stem) (tests)
 * num_steps may be 0 with noise_removal=1: one forward + argmax from x_init.
 * flags: 0, or DRK_SAMPLE_GENERIC = run the general build even where the specialised one (192 < L <= 208, dilations in
 *        {1..4, 16, 64}: immediate addressing, static live ranges) applies; both give the same bits (tests). */
#define DRK_SAMPLE_GENERIC 1
int drk_sample_bf16(const void* wstream, const void* stem_w, const void* head2_w, const float* vecs, const float* tbias, long tb_fwd_stride,
                    const float* mc_table, const float* uniforms, const int64_t* x_init, int64_t* x_out,
                    signed char* traj_out, float* logits_out, int dump_fwd, float* dbg_feat, int dbg_layer, int B, int L, int num_steps, int n_layers,
                    const int* dilations, int noise_removal, uint64_t seed, uint64_t sample_offset, int flags, void* stream);

/* ---- one residual layer per launch, bf16 MFMA (any L; BASELINE.json configs[4]: 1024-bp sequences; and the
 * differentiable bf16 path of _sample_finetune_gradient, diffusion_gosai_update.py:552-630) ----------------------------
 * models/dnaconv.py:88-97 for one layer:  feat_out = feat + relu(conv_dilated(h) + bias)   (feat NULL: no residual — the
 * head's final_conv.0, :104), and, when h_out is not NULL, the NEXT layer's input
 * h_out = bf16(LayerNorm(feat_out + tbias) * gamma + beta) (ln = 1) or bf16(feat_out) (ln = 0).
 * h, h_out: bf16 [B*L][128], distinct buffers (neighbouring workgroups read h while h_out is written);
 * feat, feat_out: float32 [B*L][128] (may alias); wfrag: this layer's taps in the fragment order of
 * drakes_amd/fastpath.py::pack_stream; tbias: [128] of batch row 0, tb_stride floats between rows (0: shared).
 * For the backward pass (all optional): relu_mask uint8 [B*L][128] = (pre-activation > 0); mean, rstd float32 [B*L] =
 * the statistics of that next LayerNorm.  relu = 0 with bias = NULL, feat = NULL, h_out = NULL is the convolution's
 * INPUT GRADIENT: h = dc (bf16), wfrag = the stream packed from weight.transpose(0,1).flip(2), feat_out = dh. */
int drk_layer_bf16(const void* h, const float* feat, float* feat_out, void* h_out, const void* wfrag, const float* bias,
                   const float* tbias, long tb_stride, const float* gamma, const float* beta, int B, int L, int taps,
                   int dilation, int ln, int relu, unsigned char* relu_mask, float* mean, float* rstd, void* stream);
/* h = bf16(LayerNorm(feat + tbias) * gamma + beta): the first layer's input (dnaconv.py:92-93); mean, rstd: optional
 * float32 [B*L] outputs for the backward pass. */
int drk_ln_fwd_bf16(const float* feat, const float* tbias, long tb_stride, const float* gamma, const float* beta, void* h,
                    float* mean, float* rstd, int B, int L, void* stream);
/* drk_stem_fwd on token input (models/dnaconv.py:80-86: one-hot, Conv1d(5 -> 128, k = 9), ReLU) and drk_ln_fwd_bf16 of
 * its output (dnaconv.py:92-93) in one launch: feat float32 [B*L][128] and h bf16 [B*L][128] are both written, with the
 * bits the two calls give; w_t [9][5][128] as for drk_stem_fwd; tbias NULL or one row per batch element (stride
 * tb_stride floats, 0 = shared); mean, rstd optional float32 [B*L]. */
int drk_stem_ln_bf16(const int64_t* x, const float* w_t, const float* bias, const float* tbias, long tb_stride,
                     const float* gamma, const float* beta, float* feat, void* h, float* mean, float* rstd, int B, int L,
                     void* stream);
/* The denoiser's head in one launch (models/dnaconv.py:60-63, 103-104: Conv1d(128 -> 128, k = 1), ReLU, Conv1d(128 -> 5,
 * k = 1)): h = bf16 of the last residual state [B*L][128]; wfrag = final_conv.0 in fragment order (the stream's last tap),
 * bias [128]; head_w [5][128], head_b [5] float32; logits float32 [B*L][5].  The 128-channel intermediate stays on the
 * chip; the bits are those of drk_layer_bf16(taps = 1, relu = 1) followed by drk_head_out_fwd. */
int drk_head_bf16(const void* h, const void* wfrag, const float* bias, const float* head_w, const float* head_b,
                  float* logits, int B, int L, void* stream);
/* SEVERAL consecutive residual layers (models/dnaconv.py:88-101, dilations dnaconv.py:53-58) in one launch, for
 * sequences beyond the fused sampler's 224 positions: layers first_layer .. first_layer + n_group - 1 applied to the
 * float32 residual stream feat_in [B*L][128] -> feat_out, every workgroup keeping a 256-row image on the chip for the
 * whole group (csrc/drk_group_bf16.hip).  Same rounding points as drk_layer_bf16 run layer by layer.
 *   stride > 1: the rows of one residue class (position mod stride) of one sequence per workgroup; every dilation a
 *               multiple of stride, dilation / stride in {1..4, 16}, ceil(L / stride) <= 256, tiles16 = 16; feat_out
 *               may be feat_in.
 *   stride = 1: consecutive rows with a recomputed halo of 4 * (sum of the dilations) rows either side; dilations in
 *               1..4; tiles16 = 14, 16 or 18 (image rows / 16); feat_out != feat_in.
 * wstream, vecs, tbias: the whole model's fragment stream, vector table and time biases [n_layers][128] (one row for
 * the batch) as drk_sample_bf16 takes them; dilations: the group's n_group true dilations.
 * In: feat_in, or (first_layer = 0, stride = 1 only) the tokens x_tok int64 [B*L] with the stem's w_t [9][5][128] and bias
 * as drk_stem_fwd takes them: the stem (dnaconv.py:80-86) then runs in the launch, with drk_stem_fwd's bits.
 * Out, any of: feat_out; h_out bf16 [B*L][128] = bf16 of the outgoing stream (the operand of drk_head_bf16); logits
 * float32 [B*L][5] with head_w [5][128], head_b [5]: the head (dnaconv.py:60-63, 103-104) runs in the launch with
 * drk_head_bf16's arithmetic — only for the group that ends the stack (the tap behind its last layer in wstream is
 * final_conv.0). */
int drk_layer_group_bf16(const float* feat_in, const int64_t* x_tok, const float* stem_w_t, const float* stem_b,
                         float* feat_out, void* h_out, float* logits, const float* head_w, const float* head_b,
                         const void* wstream, const float* vecs, const float* tbias, int B, int L, int first_layer,
                         int n_group, int stride, int tiles16, const int* dilations, void* stream);
/* Rows of feat_out one workgroup of drk_layer_group_bf16 writes at stride = 1 (0: the group does not fit the image). */
long drk_layer_group_bf16_rows(int n_group, int tiles16);

/* ---- bf16 backward of the denoiser (the differentiable bf16 path; reference: torch autograd over
 * models/dnaconv.py:88-101 inside _sample_finetune_gradient, diffusion_gosai_update.py:589-630) ----------------------
 * The convolution's INPUT gradient is drk_layer_bf16 (relu = 0) on the transposed, tap-flipped stream. */
/* dc = bf16(g * gate): g float32 [B*L][128]; gate = relu_mask (uint8 [B*L][128] from drk_layer_bf16 / drk_conv_fwd_f32) or
 * (y > 0) for a saved ReLU output y (exactly one of the two non-NULL); out bf16 [B*L][128].  colsum (optional, float32
 * [128], accumulated): column sums of the fp32 products = the conv bias gradient, reduced in a fixed order through
 * `workspace` (drk_gate_bf16_workspace_bytes() bytes, 16-byte aligned). */
long drk_gate_bf16_workspace_bytes(void);
int drk_gate_bf16(const float* g, const unsigned char* relu_mask, const float* y, void* out, float* colsum, int B, int L,
                  void* workspace, long workspace_bytes, void* stream);
/* Conv1d weight gradient from bf16 operands: dwt[t][ci][co] += sum_rows h[row + (t - taps/2)*dilation][ci] * dc[row][co],
 * h, dc bf16 [B*L][128], dwt float32 [taps][128][128]; fp32 accumulation.  Deterministic split-K as drk_conv_wgrad_f32:
 * `workspace` holds one 64-KB slab per (row chunk, tap), drk_conv_wgrad_bf16_workspace_bytes(B, L, taps) bytes. */
long drk_conv_wgrad_bf16_workspace_bytes(int B, int L, int taps);
int drk_conv_wgrad_bf16(const void* h, const void* dc, float* dwt, int B, int L, int taps, int dilation, void* workspace,
                        long workspace_bytes, void* stream);

/* ---- round 3: the bf16 backward as ONE launch per layer, parameter gradients deferred -------------------------------
 * (reference: torch autograd over models/dnaconv.py:88-101 inside the K Gumbel-softmax steps of
 * _sample_finetune_gradient, diffusion_gosai_update.py:589-630; the optimizer reads the gradients only after the whole
 * chain's backward, finetune_reward_bp.py:88-93, so weight gradients are taken over all recorded steps at once.) */
/* One residual layer's backward except its weight gradient: the convolution's input gradient (drk_layer_bf16 with
 * relu = 0 on wfrag_t, the transposed tap-flipped stream) with the LayerNorm backward, the residual add and the
 * hand-over to the layer below in its epilogue (what drk_ln_bwd does in a second launch):
 *   dh = conv_T(dc);  xhat = (feat + tbias - mean) * rstd;  dyg = dh * gamma;
 *   dx = rstd * (dyg - mean_c(dyg) - xhat * mean_c(dyg * xhat))          (ln = 1;  ln = 0: dx = dh)
 *   gout = dx + gr (gr NULL: + 0; gout may be gr);  dc_next = bf16(gout * next_mask) (both NULL: none)
 * dc, dc_next bf16 [B*L][128], distinct; feat, gr, gout float32 [B*L][128]; tbias, gamma [128] (time bias shared by
 * the batch); mean, rstd float32 [B*L] as saved by drk_layer_bf16 / drk_ln_fwd_bf16; next_mask uint8 [B*L][128].
 * partials (optional): workgroup w (128 rows each, ceil(B*L / 128) of them) writes [4][128] floats at
 * partials + w * part_stride: the column sums over its rows of dh * xhat (dgamma), dh (dbeta), dx (time bias), gated
 * gout (the lower layer's conv bias gradient); the caller adds the slabs in order (drk_slab_reduce). */
int drk_layer_bwd_bf16(const void* dc, const void* wfrag_t, const float* feat, const float* tbias, const float* gamma,
                       const float* mean, const float* rstd, const float* gr, float* gout,
                       const unsigned char* next_mask, void* dc_next, float* partials, long part_stride, int B, int L,
                       int taps, int dilation, int ln, void* stream);
/* Backward of the 128 -> 5 output conv through final_conv.0's ReLU (models/dnaconv.py:61-63) in one pass:
 * dc = bf16((dlogits x w) * (y > 0)), bf16 [B*L][128] (w = final_conv.2 [5][128], y = the saved ReLU output, float32);
 * partials (optional): workgroup w (128 rows each) writes drk_head_bwd_bf16_slab_floats() = 776 floats at
 * partials + w * part_stride: dw2 [5][128] = sum_rows dlogits[v] * y[c], db0 [128] = column sums of the fp32 gated
 * gradient (final_conv.0's bias), db2 [5] = column sums of dlogits, 3 zeros. */
long drk_head_bwd_bf16_slab_floats(void);
int drk_head_bwd_bf16(const float* dlogits, const float* y, const float* w, void* dc, float* partials, long part_stride,
                      int B, int L, void* stream);
/* drk_conv_wgrad_bf16 over n_seg recorded (h, dc) pairs of one shape in ONE launch: h_ptrs / dc_ptrs are DEVICE arrays of
 * n_seg device pointers (bf16 [B*L][128] each); dwt += the sum over all pairs.  Same kernel, same deterministic
 * split-K; workspace drk_conv_wgrad_bf16_multi_workspace_bytes(n_seg, B, L, taps) bytes. */
long drk_conv_wgrad_bf16_multi_workspace_bytes(int n_seg, int B, int L, int taps);
int drk_conv_wgrad_bf16_multi(const void* const* h_ptrs, const void* const* dc_ptrs, int n_seg, float* dwt, int B, int L,
                              int taps, int dilation, void* workspace, long workspace_bytes, void* stream);
/* out[i] += slabs[0][i] + slabs[1][i] + ... + slabs[n_slabs-1][i], in that order, i < n (n % 4 == 0, 16-byte aligned
 * pointers, slab s at slabs + s * stride): the ordered second pass of the split reductions above. */
int drk_slab_reduce(const float* slabs, int n_slabs, long stride, float* out, long n, void* stream);
/* The same sum in two launches for MANY slabs of a short vector (the fused backward's 2 B partial slabs): groups of 16
 * consecutive slabs are summed in order into tmp (drk_slab_reduce2_tmp_floats(n_slabs, n) floats, 16-byte aligned), then
 * the groups in order into out: a fixed order, so the result is reproducible bit for bit. */
long drk_slab_reduce2_tmp_floats(int n_slabs, long n);
int drk_slab_reduce2(const float* slabs, int n_slabs, long stride, float* out, long n, float* tmp, void* stream);

/* The whole differentiable denoiser forward (models/dnaconv.py:74-110 on the bf16 layer kernels) as ONE call: the launch
 * sequence drk_stem_fwd, drk_ln_fwd_bf16, n_layers x drk_layer_bf16 (ReLU gates and LayerNorm statistics kept),
 * drk_layer_bf16 (final_conv.0), drk_head_out_fwd, walked in native code — at fine-tune batch sizes the launches are
 * shorter than a Python-side sequence's host work.  Exactly one of x (int64 [B*L]) / x_soft (float32 [B*L][5]);
 * stem_w_t [9][5][128]; wstream, vecs: drakes_amd/fastpath.py::pack_stream; tbias [n_layers][128], shared by the batch;
 * dilations: HOST array [n_layers].  Outputs, one block each: feats float32 [n_layers+1][B*L][128] (residual snapshots),
 * hs bf16 [n_layers+1][B*L][128] (LayerNorm outputs; the last = bf16 of the last residual state), masks uint8
 * [n_layers][B*L][128], means / rstds float32 [n_layers][B*L], y float32 [B*L][128] (final_conv.0's ReLU output),
 * logits float32 [B*L][5].  flags: 0, or DRK_DENOISER_PER_LAYER (below). */
int drk_denoiser_fwd_bf16(const int64_t* x, const float* x_soft, const float* stem_w_t, const float* stem_b,
                          const void* wstream, const float* vecs, const float* tbias, const float* head2_w,
                          const float* head2_b, float* feats, void* hs, unsigned char* masks, float* means, float* rstds,
                          float* y, float* logits, int B, int L, int n_layers, const int* dilations, int flags, void* stream);
/* What drk_denoiser_fwd_bf16 runs for L <= 224 and <= 20 layers unless flags has DRK_DENOISER_PER_LAYER
 * (drk_denoiser_is_fused): LayerNorm 0, all residual layers and final_conv.0 in ONE launch — one sequence per workgroup,
 * the residual stream in registers, the LayerNorm outputs in LDS (drk_sample_bf16's body), the saved tensors streamed
 * out — instead of 22 launches.  In that mode two of the saved blocks change FORMAT (only the fused backward reads them;
 * every wave access is then one contiguous piece): `feats` = feat0 float32 [B*L][128] (the stem's output, row-major)
 * followed by featn, the residual snapshots after layers 0..n-1 in the kernels' lane order
 * (drk_train_fused_featn_floats(B, n_layers) floats); `masks` = maskp, the ReLU gates as one uint64 per lane and layer
 * (drk_train_fused_maskp_bytes(B, n_layers) bytes); layouts: drakes_amd/csrc/drk_train_fused_bf16.hip, and
 * drakes_amd/fastpath.py::saved_feats_rows / saved_masks_rows map them back onto rows.
 * drk_denoiser_feats_floats / drk_denoiser_masks_bytes give the sizes of the two blocks for either mode;
 * drk_denoiser_bwd_bf16 must be called with the flags the forward ran with. */
#define DRK_DENOISER_PER_LAYER 1
int drk_denoiser_is_fused(int L, int n_layers, int flags);
long drk_denoiser_feats_floats(int B, int L, int n_layers, int flags);
long drk_denoiser_masks_bytes(int B, int L, int n_layers, int flags);
long drk_train_fused_featn_floats(int B, int n_layers);
long drk_train_fused_maskp_bytes(int B, int n_layers);
int drk_train_fwd_fused_bf16(const void* wstream, const float* vecs, const float* tbias, const float* feat0, float* featn,
                             void* hs, void* maskp, float* means, float* rstds, float* y, int B, int L, int n_layers,
                             const int* dilations, void* stream);
/* Its backward except the convolutions' weight gradients and the time path (torch autograd over models/dnaconv.py:74-110
 * in the reference): drk_head_bwd_bf16, (n_layers + 1) x drk_layer_bwd_bf16, drk_relu_bwd, drk_slab_reduce,
 * drk_colsum128, drk_stem_bwd in one call.  wstream_t: drakes_amd/fastpath.py::pack_stream_T; feats, masks, means, rstds,
 * y: the forward's blocks.  dcs bf16 [dcs_slots][B*L][128]: dcs_slots = n_layers + 1 keeps every convolution's gated
 * output gradient (slot i pairs with hs[i] in drk_conv_wgrad_bf16_multi; slot n_layers = final_conv.0), dcs_slots = 2
 * ping-pongs (a frozen model; partials must be NULL).  gr: float32 [B*L][128] work buffer.  partials:
 * drk_denoiser_bwd_bf16_slabs(B, L) (the launches' own max(2 B, ceil(B*L / 128)) and the reduction's first stage)
 * slabs of drk_denoiser_bwd_bf16_slab_floats(n_layers) floats — slot i < n_layers: layer i's [4][128] of
 * drk_layer_bwd_bf16, slot n_layers: final_conv.0's input-gradient launch, then drk_head_bwd_bf16's slab — summed in
 * order into acc (+=, same length); stem_dw [9][5][128], stem_db [128] (+=, atomics); dx float32 [B*L][5] (x_soft only,
 * may be NULL).  partials and acc come together; stem_dw and stem_db come together and only with partials — a caller that
 * keeps every call's gr (= dL/d stem pre-activation on return) leaves them NULL and forms the stem's gradients once per
 * chain with drk_stem_wgrad_multi. */
long drk_denoiser_bwd_bf16_slab_floats(int n_layers);
long drk_denoiser_bwd_bf16_slabs(int B, int L);
int drk_denoiser_bwd_bf16(const float* dlogits, const float* y, const float* head2_w, const void* wstream_t,
                          const float* vecs, const float* tbias, const float* feats, const unsigned char* masks,
                          const float* means, const float* rstds, void* dcs, int dcs_slots, float* gr, float* partials,
                          float* acc, const int64_t* x, const float* x_soft, const float* stem_w_t, float* stem_dw,
                          float* stem_db, float* dx, int B, int L, int n_layers, const int* dilations, int flags, void* stream);
/* What drk_denoiser_bwd_bf16 runs between drk_head_bwd_bf16 and the stem's backward for L <= 224 and <= 20 layers unless
 * flags has DRK_DENOISER_PER_LAYER: final_conv.0's input gradient and every residual layer's backward (the arithmetic of
 * drk_layer_bwd_bf16) in ONE launch, one sequence per workgroup, the residual-gradient stream in registers, the gated
 * gradients in LDS.  feat0 / featn / maskp: the fused forward's blocks (above).  dcs slot n_layers (mod dcs_slots) is read (drk_head_bwd_bf16's output), slots n_layers-1 .. 0 are
 * written when dcs_slots = n_layers + 1; dpre float32 [B*L][128] = dL/d(stem pre-activation) (the ReLU gate of the stem
 * applied); partials: 2 B slabs (two per sequence) of part_stride >= 512 (n_layers + 1) floats, slot i at 512 i. */
int drk_train_bwd_fused_bf16(const void* wstream_t, const float* vecs, const float* tbias, const float* feat0,
                             const float* featn, const void* maskp, const float* means, const float* rstds, void* dcs,
                             int dcs_slots, float* dpre, float* partials, long part_stride, int B, int L, int n_layers,
                             const int* dilations, void* stream);

#ifdef __cplusplus
}
#endif
#endif
