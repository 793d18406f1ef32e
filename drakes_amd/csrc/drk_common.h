// Shared device helpers for the DRAKES hot-path kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRK_V 5          // vocabulary incl. mask   (diffusion_gosai_update.py:82-88)
#define DRK_MASK 4       // mask token id
#define DRK_C 128        // hidden_dim of the CNN denoiser (configs_gosai/model/dnaconv.yaml)
#define DRK_TAPS 9       // Conv1d kernel_size        (models/dnaconv.py:53-57)
#define DRK_NEG_INF (-1000000.0f)   // diffusion_gosai_update.py:127
#define DRK_MAX_DEVICES 64   // per-device set-up flags (one node holds 8 GPUs)
#define DRK_VERSION 311     // bump whenever a kernel changes: bench.py ties recorded PMC traffic to it, every bench line carries it

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// ---- Philox4x32-10 (Salmon et al., SC'11).  Counter layout: see include/drakes_hip.h --------
__device__ __forceinline__ void drk_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1, uint32_t r[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;   // one v_mad_u64_u32 each
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// Five uniforms in [0,1) on the 2^-24 grid for (seed, step, global sample, position).
__device__ __forceinline__ void drk_step_uniforms(uint64_t seed, uint32_t step, uint64_t sample,
                                                  uint32_t pos, float u[DRK_V]) {
  uint32_t r[4];
  drk_philox4x32_10(pos, (uint32_t)sample, step, 0x44524B53u ^ (uint32_t)(sample >> 32),
                    (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const float s = 5.9604644775390625e-08f;  // 2^-24
  u[0] = (float)(r[0] >> 8) * s;
  u[1] = (float)(r[1] >> 8) * s;
  u[2] = (float)(r[2] >> 8) * s;
  u[3] = (float)(r[3] >> 8) * s;
  u[4] = (float)(((r[0] & 255u) << 16) | ((r[1] & 255u) << 8) | (r[2] & 255u)) * s;
}

// ---- SUBS log-probabilities of one position (diffusion_gosai_update.py:243-253) --------------
// lg: raw backbone logits (5).  tok: held token (or the argmax of a soft state).  Writes lp[5].
__device__ __forceinline__ void drk_subs_logp(const float lg[DRK_V], int tok, float lp[DRK_V]) {
  float l[DRK_V];
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) l[v] = lg[v];
  l[DRK_MASK] = l[DRK_MASK] + DRK_NEG_INF;
  float m = l[0];
#pragma unroll
  for (int v = 1; v < DRK_V; ++v) m = fmaxf(m, l[v]);
  const float ms = (fabsf(m) == INFINITY) ? 0.0f : m;   // torch.logsumexp guards infinite maxima
  float s = 0.0f;
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) s += expf(l[v] - ms);
  const float lse = logf(s) + ms;
  if (tok != DRK_MASK) {
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) lp[v] = (v == tok) ? 0.0f : DRK_NEG_INF;
  } else {
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) lp[v] = l[v] - lse;
  }
}

// q_xs of _ddpm_update (diffusion_gosai_update.py:478-480)
__device__ __forceinline__ void drk_posterior(const float lp[DRK_V], float mc_t, float mc_s, float q[DRK_V]) {
  const float d = mc_t - mc_s;
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) q[v] = expf(lp[v]) * d;
  q[DRK_MASK] = mc_s;
}

// _sample_categorical (diffusion_gosai_update.py:24-28): divide, then first-index argmax.
__device__ __forceinline__ int drk_gumbel_argmax(const float q[DRK_V], const float u[DRK_V]) {
  int best = 0;
  float bv = 0.0f;
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) {
    const float g = 1e-10f - logf(u[v] + 1e-10f);
    const float r = q[v] / g;
    if (v == 0 || r > bv) { bv = r; best = v; }
  }
  return best;
}

#define DRK_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return drk_set_error(e__, __func__); \
  } while (0)

// Shape guard of every entry point: a negative batch or empty length is an error, an EMPTY batch
// is a successful no-op (its pointers may be null), checked before the pointers are.
#define DRK_SHAPE_GUARD(name)                                                        \
  if (B < 0 || L <= 0) return drk_set_error_msg(DRK_EINVAL, name ": bad shape");     \
  if (B == 0) return DRK_OK

// out[i] += slabs[0][i] + slabs[1][i] + ... + slabs[n_slabs-1][i] in that order, i < n (n % 4 == 0, every pointer
// 16-byte aligned); slab s starts at slabs + s * stride.  The ordered second pass of every split-K reduction here
// (defined in drk_train_f32.hip).
int drk_slab_reduce_launch(const float* slabs, int n_slabs, long stride, float* out, long n, hipStream_t stream);

int drk_set_error(hipError_t e, const char* where);
int drk_set_error_msg(int code, const char* msg);
