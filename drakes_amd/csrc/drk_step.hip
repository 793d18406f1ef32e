// Reverse-step kernels: SUBS parameterisation, absorbing-diffusion posterior, Gumbel-argmax,
// carry-over unmasking.  HBM-bound elementwise work: one lane per sequence position, 5 logits
// (20 B) + token (8 B) in, token (8 B) out; uniforms either read (20 B) or generated in-register
// by Philox.  Logits of consecutive lanes are contiguous, so every wave-load is a dense
// 1280-byte span; tokens are dense 512-byte spans.
#include "drk_common.h"
#include "../../include/drakes_hip.h"

#include <stdio.h>
#include <string.h>

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[256] = "";

int drk_set_error(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return DRK_EHIP;
}
int drk_set_error_msg(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
extern "C" const char* drk_last_error(void) { return g_err; }
extern "C" int drk_version(void) { return DRK_VERSION; }

// ------------------------------------------------------------------------------------ kernels
__device__ __forceinline__ void load5(const float* __restrict__ p, long i, float v[DRK_V]) {
#pragma unroll
  for (int k = 0; k < DRK_V; ++k) v[k] = p[i * DRK_V + k];
}

__device__ __forceinline__ int soft_argmax(const float v[DRK_V]) {
  int best = 0;
#pragma unroll
  for (int k = 1; k < DRK_V; ++k)
    if (v[k] > v[best]) best = k;
  return best;
}

__global__ void __launch_bounds__(256) k_step_uniforms(float* __restrict__ u, long n, int L, uint64_t seed,
                                                       uint32_t step, uint64_t sample_offset) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float uu[DRK_V];
    const uint32_t row = (uint32_t)i / (uint32_t)L;   // n < 2^32, checked by the entry point
    drk_step_uniforms(seed, step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, uu);
#pragma unroll
    for (int k = 0; k < DRK_V; ++k) u[i * DRK_V + k] = uu[k];
  }
}

__global__ void __launch_bounds__(256) k_subs_fwd(const float* __restrict__ logits, const int64_t* __restrict__ x,
                                                  const float* __restrict__ x_soft, float* __restrict__ logp, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float lg[DRK_V], lp[DRK_V];
    load5(logits, i, lg);
    int tok;
    if (x) {
      tok = (int)x[i];
    } else {
      float xs[DRK_V];
      load5(x_soft, i, xs);
      tok = soft_argmax(xs);
    }
    drk_subs_logp(lg, tok, lp);
#pragma unroll
    for (int k = 0; k < DRK_V; ++k) logp[i * DRK_V + k] = lp[k];
  }
}

template <bool INJECT>
__global__ void __launch_bounds__(256)
    k_reverse_step_hard(const float* __restrict__ logits, const int64_t* x,  // x_next may alias x
                        const float* __restrict__ uniforms, int64_t* x_next, long n, int L, float mc_t,
                        float mc_s, const float* __restrict__ mc_rows, uint64_t seed, uint32_t step,
                        uint64_t sample_offset) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tok = (int)x[i];
    int out = tok;
    if (tok == DRK_MASK) {  // carry-over: an unmasked position keeps its token (:482-486)
      float lg[DRK_V], lp[DRK_V], q[DRK_V], u[DRK_V];
      load5(logits, i, lg);
      // n < 2^32 on every call the entry point accepts (B, L are ints): 32-bit divide, not the 64-bit one
      const uint32_t row = (uint32_t)i / (uint32_t)L;
      if (INJECT)
        load5(uniforms, i, u);
      else
        drk_step_uniforms(seed, step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, u);
      if (mc_rows) {   // per-row timesteps (t [B,1] with unequal rows, :462-474)
        mc_t = mc_rows[2 * row];
        mc_s = mc_rows[2 * row + 1];
      }
      drk_subs_logp(lg, DRK_MASK, lp);
      drk_posterior(lp, mc_t, mc_s, q);
      out = drk_gumbel_argmax(q, u);
    }
    x_next[i] = out;
  }
}

__global__ void __launch_bounds__(256) k_denoise_argmax(const float* __restrict__ logits, const int64_t* __restrict__ x,
                                                        int64_t* __restrict__ x_out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tok = (int)x[i];
    float lg[DRK_V], lp[DRK_V];
    load5(logits, i, lg);
    drk_subs_logp(lg, tok, lp);
    int best = 0;
#pragma unroll
    for (int k = 1; k < DRK_V - 1; ++k)
      if (lp[k] > lp[best]) best = k;
    x_out[i] = best;
  }
}

// Forward noising q(x_t | x_0) of the absorbing process (diffusion_gosai_update.py:425-436): a position becomes the
// mask token with probability move_chance.  ONE uniform per position (the reference draws torch.rand(*x.shape)): the
// first of the step's Philox uniforms, or uniforms[i] when injected.
template <bool INJECT>
__global__ void __launch_bounds__(256) k_q_xt(const int64_t* __restrict__ x0, const float* __restrict__ uniforms,
                                              int64_t* __restrict__ xt, long n, int L, float move_chance,
                                              const float* __restrict__ mc_rows, uint64_t seed, uint32_t step,
                                              uint64_t sample_offset) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float u;
    const uint32_t row = (uint32_t)i / (uint32_t)L;
    if (INJECT) {
      u = uniforms[i];
    } else {
      float uu[DRK_V];
      drk_step_uniforms(seed, step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, uu);
      u = uu[0];
    }
    const float mc = mc_rows ? mc_rows[row] : move_chance;   // per-row move chance [B] (:425-436 takes [B,1])
    xt[i] = (u < mc) ? (int64_t)DRK_MASK : x0[i];
  }
}

// One term of get_likelihood (diffusion_gosai_update.py:911-917): acc[b,l] += multiplier * log p(x0[b,l] | x_t), with
// the SUBS log-probability taken from raw logits (0 where x_t still shows the token).
__global__ void __launch_bounds__(256) k_likelihood_accum(const float* __restrict__ logits, const int64_t* __restrict__ x0,
                                                          const int64_t* __restrict__ xt, float multiplier,
                                                          float* __restrict__ acc, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float lg[DRK_V], lp[DRK_V];
    load5(logits, i, lg);
    drk_subs_logp(lg, (int)xt[i], lp);
    const int tok = (int)x0[i];
    float v = lp[0];
#pragma unroll
    for (int k = 1; k < DRK_V; ++k)
      if (tok == k) v = lp[k];
    acc[i] += v * multiplier;
  }
}

// Graph-replayable form of the hard step: everything that changes from step to step comes from device memory, so ONE
// captured launch sequence (hipGraph) serves every step and every call.  params[k] = (mc_t, mc_s) of step k; rng =
// (seed, sample_offset); *counter = the current
// step, advanced by k_step_advance (a separate 1-thread node AFTER the step kernel: no block of the step kernel can
// observe the increment).
__global__ void __launch_bounds__(256)
    k_reverse_step_hard_table(const float* __restrict__ logits, const int64_t* x, int64_t* x_next, long n, int L,
                              const float* __restrict__ params, const int* __restrict__ counter,
                              const uint64_t* __restrict__ rng) {
  const int step = *counter;
  const uint64_t seed = rng[0], sample_offset = rng[1];
  const float mc_t = params[2 * step], mc_s = params[2 * step + 1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int tok = (int)x[i];
    int out = tok;
    if (tok == DRK_MASK) {
      float lg[DRK_V], lp[DRK_V], q[DRK_V], u[DRK_V];
      load5(logits, i, lg);
      const uint32_t row = (uint32_t)i / (uint32_t)L;
      drk_step_uniforms(seed, (uint32_t)step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, u);
      drk_subs_logp(lg, DRK_MASK, lp);
      drk_posterior(lp, mc_t, mc_s, q);
      out = drk_gumbel_argmax(q, u);
    }
    x_next[i] = out;
  }
}

__global__ void k_step_advance(int* counter) { *counter += 1; }

// ---- wide vocabulary (protein inverse folding, V = 22): Interpolant.sample, drakes_protein/fmif/fm_utils.py:113-134 ----
// Same absorbing-diffusion step with a sampling temperature: logp = logits/temp - logsumexp(logits/temp) with the mask
// logit at -1e6, q = exp(logp) * d_t, q[mask] = move_chance_s, Gumbel-argmax, carry-over; plus the clean prediction
// argmax over the non-mask logits.  One lane per position; its V logits are V*4 contiguous bytes.
#define DRK_VMAX 32
__device__ __forceinline__ void drk_step_uniforms_wide(uint64_t seed, uint32_t step, uint64_t sample, uint32_t pos,
                                                       int nclass, float* u) {
  const float sc = 5.9604644775390625e-08f;  // 2^-24
  for (int blk = 0; 4 * blk < nclass; ++blk) {
    uint32_t r[4];
    drk_philox4x32_10(pos, (uint32_t)sample, step, 0x50524F54u ^ (uint32_t)(sample >> 32) ^ ((uint32_t)blk << 24),
                      (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (4 * blk + w < nclass) u[4 * blk + w] = (float)(r[w] >> 8) * sc;
  }
}

template <int V, bool INJECT>
__global__ void __launch_bounds__(256)
    k_reverse_step_hard_wide(const float* __restrict__ logits, const int64_t* x, const float* __restrict__ uniforms,
                             int64_t* x_next, int64_t* __restrict__ pred_clean, long n, int L, int mask_index, float temp,
                             float d_t, float mc_s, uint64_t seed, uint32_t step, uint64_t sample_offset) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float lg[V];
#pragma unroll
    for (int v = 0; v < V; ++v) lg[v] = logits[i * V + v];
    if (pred_clean) {   // argmax over the logits with the mask column at -1e9 (first index wins, :115-117)
      int best = 0;
      float bv = (0 == mask_index) ? -1e9f : lg[0];
#pragma unroll
      for (int v = 1; v < V; ++v) {
        const float c = (v == mask_index) ? -1e9f : lg[v];
        if (c > bv) { bv = c; best = v; }
      }
      pred_clean[i] = best;
    }
    const int tok = (int)x[i];
    int out = tok;
    if (tok == mask_index) {
      float m = -INFINITY;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        lg[v] = ((v == mask_index) ? DRK_NEG_INF : lg[v]) / temp;
        m = fmaxf(m, lg[v]);
      }
      const float ms = (fabsf(m) == INFINITY) ? 0.0f : m;
      float s = 0.0f;
#pragma unroll
      for (int v = 0; v < V; ++v) s += expf(lg[v] - ms);
      const float lse = logf(s) + ms;
      float u[V];
      if (INJECT) {
#pragma unroll
        for (int v = 0; v < V; ++v) u[v] = uniforms[i * V + v];
      } else {
        const uint32_t row = (uint32_t)i / (uint32_t)L;
        drk_step_uniforms_wide(seed, step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, V, u);
      }
      float bv = 0.0f;
      int best = 0;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const float q = (v == mask_index) ? mc_s : expf(lg[v] - lse) * d_t;
        const float r = q / (1e-10f - logf(u[v] + 1e-10f));
        if (v == 0 || r > bv) { bv = r; best = v; }
      }
      out = best;
    }
    x_next[i] = out;
  }
}

// ------------------------------------------------------------------------------------ C ABI
static inline int grid_for(long n) {
  long g = (n + 255) / 256;
  if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride beyond
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int drk_step_uniforms(float* u, int B, int L, uint64_t seed, uint32_t step, uint64_t sample_offset,
                                 void* stream) {
  DRK_SHAPE_GUARD("drk_step_uniforms");
  if (!u || B < 0 || L <= 0) return drk_set_error_msg(DRK_EINVAL, "drk_step_uniforms: bad arguments");
  const long n = (long)B * L;
  if (n == 0) return DRK_OK;
  if (n >= (1L << 32)) return drk_set_error_msg(DRK_EINVAL, "drk_step_uniforms: B*L must be below 2^32");
  hipLaunchKernelGGL(k_step_uniforms, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, n, L, seed, step,
                     sample_offset);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_subs_fwd(const float* logits, const int64_t* x, const float* x_soft, float* logp, int B, int L,
                            void* stream) {
  DRK_SHAPE_GUARD("drk_subs_fwd");
  if (!logits || !logp || (!x && !x_soft) || B < 0 || L <= 0)
    return drk_set_error_msg(DRK_EINVAL, "drk_subs_fwd: bad arguments");
  const long n = (long)B * L;
  if (n == 0) return DRK_OK;
  hipLaunchKernelGGL(k_subs_fwd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits, x, x_soft, logp, n);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_reverse_step_hard(const float* logits, const int64_t* x, const float* uniforms, int64_t* x_next,
                                     int B, int L, float mc_t, float mc_s, const float* mc_rows, uint64_t seed,
                                     uint32_t step, uint64_t sample_offset, void* stream) {
  DRK_SHAPE_GUARD("drk_reverse_step_hard");
  if (!logits || !x || !x_next || B < 0 || L <= 0)
    return drk_set_error_msg(DRK_EINVAL, "drk_reverse_step_hard: bad arguments");
  const long n = (long)B * L;
  if (n == 0) return DRK_OK;
  if (n >= (1L << 32)) return drk_set_error_msg(DRK_EINVAL, "drk_reverse_step_hard: B*L must be below 2^32");
  if (uniforms)
    hipLaunchKernelGGL(k_reverse_step_hard<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits, x,
                       uniforms, x_next, n, L, mc_t, mc_s, mc_rows, seed, step, sample_offset);
  else
    hipLaunchKernelGGL(k_reverse_step_hard<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits, x,
                       uniforms, x_next, n, L, mc_t, mc_s, mc_rows, seed, step, sample_offset);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_denoise_argmax(const float* logits, const int64_t* x, int64_t* x_out, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_denoise_argmax");
  if (!logits || !x || !x_out || B < 0 || L <= 0)
    return drk_set_error_msg(DRK_EINVAL, "drk_denoise_argmax: bad arguments");
  const long n = (long)B * L;
  if (n == 0) return DRK_OK;
  hipLaunchKernelGGL(k_denoise_argmax, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits, x, x_out, n);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_q_xt(const int64_t* x0, const float* uniforms, int64_t* xt, int B, int L, float move_chance,
                        const float* mc_rows, uint64_t seed, uint32_t step, uint64_t sample_offset, void* stream) {
  DRK_SHAPE_GUARD("drk_q_xt");
  if (!x0 || !xt) return drk_set_error_msg(DRK_EINVAL, "drk_q_xt: bad arguments");
  const long n = (long)B * L;
  if (n >= (1L << 32)) return drk_set_error_msg(DRK_EINVAL, "drk_q_xt: B*L must be below 2^32");
  if (uniforms)
    hipLaunchKernelGGL(k_q_xt<true>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x0, uniforms, xt, n, L,
                       move_chance, mc_rows, seed, step, sample_offset);
  else
    hipLaunchKernelGGL(k_q_xt<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x0, uniforms, xt, n, L,
                       move_chance, mc_rows, seed, step, sample_offset);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_likelihood_accum(const float* logits, const int64_t* x0, const int64_t* xt, float multiplier,
                                    float* acc, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_likelihood_accum");
  if (!logits || !x0 || !xt || !acc) return drk_set_error_msg(DRK_EINVAL, "drk_likelihood_accum: bad arguments");
  const long n = (long)B * L;
  hipLaunchKernelGGL(k_likelihood_accum, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits, x0, xt, multiplier,
                     acc, n);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_reverse_step_hard_table(const float* logits, const int64_t* x, int64_t* x_next, int B, int L,
                                           const float* mc_table, int* step_counter, const uint64_t* rng,
                                           void* stream) {
  DRK_SHAPE_GUARD("drk_reverse_step_hard_table");
  if (!logits || !x || !x_next || !mc_table || !step_counter || !rng)
    return drk_set_error_msg(DRK_EINVAL, "drk_reverse_step_hard_table: bad arguments");
  const long n = (long)B * L;
  if (n >= (1L << 32)) return drk_set_error_msg(DRK_EINVAL, "drk_reverse_step_hard_table: B*L must be below 2^32");
  hipLaunchKernelGGL(k_reverse_step_hard_table, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits, x, x_next, n,
                     L, mc_table, step_counter, rng);
  DRK_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_reverse_step_hard_wide(const float* logits, const int64_t* x, const float* uniforms, int64_t* x_next,
                                          int64_t* pred_clean, int B, int L, int V, int mask_index, float temp, float d_t,
                                          float move_chance_s, uint64_t seed, uint32_t step, uint64_t sample_offset,
                                          void* stream) {
  DRK_SHAPE_GUARD("drk_reverse_step_hard_wide");
  if (!logits || !x || !x_next || temp <= 0.f || mask_index < 0 || mask_index >= V)
    return drk_set_error_msg(DRK_EINVAL, "drk_reverse_step_hard_wide: bad arguments");
  if (V != 22) return drk_set_error_msg(DRK_EINVAL, "drk_reverse_step_hard_wide: built for V = 22 (20 amino acids, X, mask)");
  const long n = (long)B * L;
  if (n >= (1L << 32)) return drk_set_error_msg(DRK_EINVAL, "drk_reverse_step_hard_wide: B*L must be below 2^32");
  if (uniforms)
    hipLaunchKernelGGL((k_reverse_step_hard_wide<22, true>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits, x,
                       uniforms, x_next, pred_clean, n, L, mask_index, temp, d_t, move_chance_s, seed, step, sample_offset);
  else
    hipLaunchKernelGGL((k_reverse_step_hard_wide<22, false>), dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logits, x,
                       uniforms, x_next, pred_clean, n, L, mask_index, temp, d_t, move_chance_s, seed, step, sample_offset);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
