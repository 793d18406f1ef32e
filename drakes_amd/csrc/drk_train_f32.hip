// Backward kernels of the fp32 denoiser and of the soft (Gumbel-softmax) reverse step, for the
// truncated reward backpropagation of Diffusion._sample_finetune_gradient
// (diffusion_gosai_update.py:552-630) and the KL loop of fine_tune (finetune_reward_bp.py:60-75).
// The reference gets all of this from torch autograd; here each backward op is a kernel and
// drakes_amd/autograd.py chains them.  Parameter gradients are ACCUMULATED with fp32 atomics into
// buffers the caller zeroes (float atomic sums depend on arrival order: last-bit differences
// between runs are expected and covered by the stated gradient tolerance).
#include "drk_common.h"
#include "../../include/drakes_hip.h"

#include <atomic>

#define ARGCHECK(cond, name) \
  if (!(cond)) return drk_set_error_msg(DRK_EINVAL, name ": bad arguments")

static inline int grid_for(long n, int per_block) {
  long g = (n + per_block - 1) / per_block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------ elementwise
// out = g * mask   (mask: uint8 0/1 written by drk_conv_fwd_f32)      -- ReLU backward
__global__ void __launch_bounds__(256) k_mask_mul(const float* __restrict__ g, const unsigned char* __restrict__ mask,
                                                  float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)g)[i];
    const uchar4 m = ((const uchar4*)mask)[i];
    ((f32x4*)out)[i] = f32x4{m.x ? v[0] : 0.f, m.y ? v[1] : 0.f, m.z ? v[2] : 0.f, m.w ? v[3] : 0.f};
  }
}
// out = g * (y > 0)   (y: a saved ReLU OUTPUT)
__global__ void __launch_bounds__(256) k_relu_bwd(const float* __restrict__ g, const float* __restrict__ y,
                                                  float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)g)[i], yy = ((const f32x4*)y)[i];
    ((f32x4*)out)[i] = f32x4{yy[0] > 0.f ? v[0] : 0.f, yy[1] > 0.f ? v[1] : 0.f, yy[2] > 0.f ? v[2] : 0.f,
                             yy[3] > 0.f ? v[3] : 0.f};
  }
}

// Rows per workgroup of the kernels below that end in one float atomic per column and workgroup (k_colsum128,
// k_head_out_bwd; k_stem_wgrad, 45 multiply-adds per row and thread, is faster with 64): atomics on the same few hundred addresses serialise, so fewer, taller workgroups win at
// fine-tune batch sizes (25 600 rows: 200 workgroups of 128 rows instead of 400 of 64).
#define CS_ROWS 128
// ------------------------------------------------------------------------------------ column sums
// out[seg][c] += sum over the rows of segment seg of x[row][c];  rows = nseg * seg_rows, 128 columns.
// Workgroup = 256 threads = 2 row-phases x 128 columns over a block of CS_ROWS rows of one segment.
__global__ void __launch_bounds__(256) k_colsum128(const float* __restrict__ x, float* __restrict__ out, long seg_rows) {
  __shared__ float red[128];
  const int c = threadIdx.x & 127, ph = threadIdx.x >> 7;
  const long seg = blockIdx.y;
  const long r0 = (long)blockIdx.x * CS_ROWS;
  float s = 0.f;
  for (long rb = r0 + ph; rb < r0 + CS_ROWS; rb += 16) {      // 8 rows in flight per thread (row by row: one round trip each)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long r = rb + 2 * j;
      v[j] = r < seg_rows ? x[(seg * seg_rows + r) * DRK_C + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  if (ph == 1) red[c] = s;
  __syncthreads();
  if (ph == 0) atomicAdd(out + seg * DRK_C + c, s + red[c]);
}

// ------------------------------------------------------------------------------------ LayerNorm backward
// Forward: xin = feat + tbias; xhat = (xin - mean) * rstd; h = xhat * gamma + beta.
// Given dh: dxin = rstd * (dyg - mean_c(dyg) - xhat * mean_c(dyg * xhat)),  dyg = dh * gamma;
//           dgamma += sum_rows dh * xhat;  dbeta += sum_rows dh.
// dfeat_out = dxin + g_resid (the gradient arriving through the residual branch); dxin is also stored
// on its own for the time-bias gradient.  32 lanes per row; per-workgroup partial column sums in LDS,
// then one atomic per column per workgroup.
#ifndef LNB_UNR
#define LNB_UNR 4
#endif
// LNB_UNR: rows per half-wave and pass: all their loads are requested before the first use
#define LNB_SLOTS 16                     // half-waves per workgroup (512 threads)
#define LNB_PASS_ROWS (LNB_SLOTS * LNB_UNR)
// `passes` passes of 64 rows per workgroup: the host sizes it so that the launch is about one workgroup per CU — the
// column sums end in one float atomic per column and workgroup on the SAME 128 addresses, and those serialise (400
// workgroups of 64 rows: 24 us at 25 600 rows, most of it the atomics; 800 of 32 rows: 29 us).
__global__ void __launch_bounds__(512) k_ln_bwd(const float* __restrict__ dh, const float* __restrict__ feat,
                                                const float* __restrict__ tbias, long tb_stride,
                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, const float* __restrict__ g_resid,
                                                float* __restrict__ dxin, float* __restrict__ dfeat,
                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                float* __restrict__ dtb, const unsigned char* __restrict__ next_mask,
                                                char* __restrict__ dc_next, float* __restrict__ dbias_next, long M, int L,
                                                int width, int passes) {
  __shared__ float pg[LNB_SLOTS][128], pb[LNB_SLOTS][128], px[LNB_SLOTS][128], pc[LNB_SLOTS][128];
  const float inv = 1.0f / (float)width;   // width < 128: statistics over the first `width` channels (drk_ln_fwd)
  const int lr = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const f32x4 gm = *(const f32x4*)(gamma + 4 * lr);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 ag = zero4, ab = zero4, ax = zero4, ac = zero4;
  for (int pass = 0; pass < passes; ++pass) {
    const long base = ((long)blockIdx.x * passes + pass) * LNB_PASS_ROWS + sub;
    if (base - sub >= M) break;
    f32x4 xv[LNB_UNR], dv[LNB_UNR], gv[LNB_UNR];
    float muv[LNB_UNR], rsv[LNB_UNR];
    unsigned mkv[LNB_UNR];
#pragma unroll
    for (int i = 0; i < LNB_UNR; ++i) {
      const long row = base + LNB_SLOTS * i;
      const long r = row < M ? row : M - 1;
      xv[i] = *(const f32x4*)(feat + r * DRK_C + 4 * lr);
      if (tbias) xv[i] += *(const f32x4*)(tbias + (r / L) * tb_stride + 4 * lr);
      muv[i] = mean[r];
      rsv[i] = rstd[r];
      dv[i] = *(const f32x4*)(dh + r * DRK_C + 4 * lr);
      gv[i] = g_resid ? *(const f32x4*)(g_resid + r * DRK_C + 4 * lr) : zero4;
      mkv[i] = next_mask ? *(const unsigned*)(next_mask + r * DRK_C + 4 * lr) : 0u;
    }
#pragma unroll
    for (int i = 0; i < LNB_UNR; ++i) {
      const long row = base + LNB_SLOTS * i;
      if (row >= M) continue;               // (uniform over the half-wave: the shuffles below stay within it)
      const float rs = rsv[i];
      const f32x4 xh = (xv[i] - muv[i]) * rs;
      const f32x4 d = dv[i];
      const f32x4 dyg = d * gm;
      float s1 = (dyg[0] + dyg[1]) + (dyg[2] + dyg[3]);
      float s2 = (dyg[0] * xh[0] + dyg[1] * xh[1]) + (dyg[2] * xh[2] + dyg[3] * xh[3]);
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
      }
      const float m1 = s1 * inv, m2 = s2 * inv;
      f32x4 dx = (dyg - m1 - xh * m2) * rs;
      if (4 * lr >= width) dx = zero4;   // padding lanes carry no gradient
      if (dxin) *(f32x4*)(dxin + row * DRK_C + 4 * lr) = dx;
      const f32x4 o = dx + gv[i];
      *(f32x4*)(dfeat + row * DRK_C + 4 * lr) = o;
      if (next_mask) {   // the layer below: dc = bf16(dfeat * its ReLU mask), and the column sums = its conv bias gradient
        const unsigned m = mkv[i];
        const f32x4 v = {(m & 0xffu) ? o[0] : 0.f, (m & 0xff00u) ? o[1] : 0.f, (m & 0xff0000u) ? o[2] : 0.f, (m & 0xff000000u) ? o[3] : 0.f};
        typedef __bf16 bf16v4_t __attribute__((ext_vector_type(4)));
        bf16v4_t pk;
        pk[0] = (__bf16)v[0]; pk[1] = (__bf16)v[1]; pk[2] = (__bf16)v[2]; pk[3] = (__bf16)v[3];
        *(bf16v4_t*)(dc_next + row * 256 + 8 * lr) = pk;
        ac += v;
      }
      ag += d * xh;
      ab += d;
      ax += dx;
    }
  }
  *(f32x4*)(&pg[sub][4 * lr]) = ag;
  *(f32x4*)(&pb[sub][4 * lr]) = ab;
  *(f32x4*)(&px[sub][4 * lr]) = ax;
  *(f32x4*)(&pc[sub][4 * lr]) = ac;
  __syncthreads();
  // four column sums x 128 columns = the 512 threads, one atomic each
  const int col = threadIdx.x & 127, which = threadIdx.x >> 7;
  float* const dst = which == 0 ? dgamma : which == 1 ? dbeta : which == 2 ? dtb : dbias_next;
  if (dst) {
    const float (*src)[128] = which == 0 ? pg : which == 1 ? pb : which == 2 ? px : pc;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < LNB_SLOTS; ++k) sum += src[k][col];
    atomicAdd(dst + col, sum);
  }
}

// ------------------------------------------------------------------------------------ conv weight gradient
// dwt[t][ci][co] += sum_rows h[row + off_t][ci] * dc[row][co]     (rows whose shifted position leaves [0,L) add 0)
// GEMM with K = rows on the f32-input MFMA: A = h^T [32 ci x 2 rows], B = dc [2 rows x 32 co]; both operands are
// row-contiguous, so every wave-load is two dense 128-byte spans.  Grid = (row chunks, taps); wave w = ci tile w, all
// four co tiles; a workgroup reduces `wg_rows` rows (chosen by the host so that small batches still fill the chip).
// DETERMINISTIC split-K: the workgroup STORES its 128x128 partial into slab (chunk, tap) of the caller's workspace
// (plain 128-byte row segments), and k_wgrad_reduce adds the slabs in chunk order: the same bits on every run, and
// plain stores move at 4-5x the rate of float atomics (MI355X_MICROARCH.md, Global float atomics).
#define WG_SLAB (DRK_C * DRK_C)   // floats per (chunk, tap) slab
#ifndef WG_DEPTH
#define WG_DEPTH 4                // steps of operands in flight + 1 (the loop is unrolled by it)
#endif
__global__ void __launch_bounds__(256) k_conv_wgrad(const float* __restrict__ h, const float* __restrict__ dc,
                                                    float* __restrict__ slabs, float* __restrict__ bslabs, long M, int L, int taps, int dil,
                                                    int wg_rows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = lane >> 5, lr = lane & 31;
  const int t = blockIdx.y;
  const int off = (t - taps / 2) * dil;
  const long r0 = (long)blockIdx.x * wg_rows;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const long rend = (r0 + wg_rows < M) ? r0 + wg_rows : M;
  int p = (int)((r0 + half) % L);                      // position of this lane's first row; advanced by 2 per step
  const float* hp = h + (r0 + half + off) * DRK_C + 32 * w + lr;
  const float* dp = dc + (r0 + half) * DRK_C + lr;
  const int nsteps = (int)((rend - r0 + 1) >> 1);     // both half-waves run the same trip count
  // Operands of the next WG_DEPTH - 1 steps are in flight under the MFMAs of a step.  The loads are UNCONDITIONAL (a step
  // past the end, or a shifted row outside its sequence, re-reads the chunk's first row) and blanked at their use: behind
  // a branch the compiler cannot count them and waited for every load in flight at each group's first MFMA — one whole
  // round trip per four steps.
  const float* const hsafe = h + r0 * DRK_C + 32 * w + lr;
  const float* const dsafe = dc + r0 * DRK_C + lr;
  auto fetch = [&](int s, float& a, f32x4& bq, int& fl) {
    const bool vb = s < nsteps && r0 + half + 2 * s < rend;
    const int q = p + off;
    const bool va = vb && q >= 0 && q < L;
    fl = (va ? 1 : 0) | (vb ? 2 : 0);
    const float* hq = va ? hp : hsafe;
    const float* dq = vb ? dp : dsafe;
    a = *hq;
    bq = f32x4{dq[0], dq[32], dq[64], dq[96]};
    hp += 2 * DRK_C;
    dp += 2 * DRK_C;
    p += 2;
    if (p >= L) p -= L;
    if (p >= L) p -= L;
  };
  float av[WG_DEPTH];
  f32x4 bv[WG_DEPTH];
  int fv[WG_DEPTH];
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};   // column sums of dc = the conv bias gradient (one wave of the centre-tap workgroup)
  const bool do_bias = bslabs && off == 0 && t == taps / 2 && w == 0;
#pragma unroll
  for (int k = 0; k < WG_DEPTH - 1; ++k) fetch(k, av[k], bv[k], fv[k]);
  for (int s = 0; s < nsteps; s += WG_DEPTH) {
#pragma unroll
    for (int k = 0; k < WG_DEPTH; ++k) {
      constexpr int dummy = 0; (void)dummy;
      fetch(s + k + WG_DEPTH - 1, av[(k + WG_DEPTH - 1) % WG_DEPTH], bv[(k + WG_DEPTH - 1) % WG_DEPTH], fv[(k + WG_DEPTH - 1) % WG_DEPTH]);
      const float au = (fv[k] & 1) ? av[k] : 0.f;
      const f32x4 bu = (fv[k] & 2) ? bv[k] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(au, bu[j], acc[j], 0, 0, 0);
      if (do_bias) bsum += bu;
    }
  }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = bsum[j] + __shfl_xor(bsum[j], 32);      // even + odd rows
      if (half == 0) bslabs[(long)blockIdx.x * DRK_C + 32 * j + lr] = v;
    }
  }
  float* dst = slabs + ((long)blockIdx.x * taps + t) * WG_SLAB;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ci = 32 * w + (i & 3) + 8 * (i >> 2) + 4 * half;
      dst[ci * DRK_C + 32 * j + lr] = acc[j][i];
    }
}

// out[i] += slabs[0][i] + slabs[1][i] + ... (fixed order), i < n (n % 4 == 0); slab s starts at slabs + s * stride.
// One f32x4 per lane, consecutive lanes on consecutive 16-byte pieces: every wave-load is a dense 1-KiB span.
__global__ void __launch_bounds__(256) k_slab_reduce(const float* __restrict__ slabs, int n_slabs, long stride,
                                                     float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < n_slabs; k0 += 8) {   // 8 independent loads in flight, then added in slab order
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k0 + j < n_slabs) v[j] = ((const f32x4*)(slabs + (long)(k0 + j) * stride))[i];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    ((f32x4*)out)[i] += s;
  }
}

// ------------------------------------------------------------------------------------ head (128 -> 5) backward
// dy[row][c] = sum_v dlogits[row][v] * w[v][c];  dw[v][c] += sum_rows dlogits[row][v] * y[row][c];  db[v] += sum_rows dlogits
__global__ void __launch_bounds__(256) k_head_out_bwd(const float* __restrict__ dlogits, const float* __restrict__ y,
                                                      const float* __restrict__ w, float* __restrict__ dy,
                                                      float* __restrict__ dw, float* __restrict__ db, long M) {
  __shared__ float ws[DRK_V * DRK_C];
  __shared__ float pw[2][DRK_V][DRK_C];
  for (int i = threadIdx.x; i < DRK_V * DRK_C; i += 256) ws[i] = w[i];
  __syncthreads();
  const int c = threadIdx.x & 127, ph = threadIdx.x >> 7;
  float aw[DRK_V] = {0.f, 0.f, 0.f, 0.f, 0.f}, ab[DRK_V] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (long rb = (long)blockIdx.x * CS_ROWS + ph; rb < (long)blockIdx.x * CS_ROWS + CS_ROWS; rb += 16) {
    // 8 rows per thread in flight (row by row the 32 dependent round trips of a thread were the whole 73 us)
    float dl[8][DRK_V], yy[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long row = rb + 2 * j;
      const long r = row < M ? row : M - 1;
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) dl[j][v] = dlogits[r * DRK_V + v];
      yy[j] = y[r * DRK_C + c];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long row = rb + 2 * j;
      if (row >= M) continue;
      float o = 0.f;
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) {
        o += dl[j][v] * ws[v * DRK_C + c];
        aw[v] += dl[j][v] * yy[j];
        ab[v] += dl[j][v];
      }
      dy[row * DRK_C + c] = o;
    }
  }
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) pw[ph][v][c] = aw[v];
  __syncthreads();
  if (ph == 0) {
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) atomicAdd(dw + v * DRK_C + c, pw[0][v][c] + pw[1][v][c]);
  }
  if (c == 0) {
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) atomicAdd(db + v, ab[v]);
  }
}

// ------------------------------------------------------------------------------------ stem backward
// dpre[row][co] = dfeat0[row][co] * (feat0 > 0) is formed by the caller (k_relu_bwd).
// dw_t[t][v][co] += sum_rows dpre[row][co] * xs[row + t - 4][v];   db via k_colsum128;
// dx[row][v] = sum_t sum_co dpre[row - (t-4)][co] * w_t[t][v][co]   (soft input only).
__global__ void __launch_bounds__(256) k_stem_wgrad(const float* __restrict__ dpre, const int64_t* __restrict__ x,
                                                    const float* __restrict__ x_soft, float* __restrict__ dw_t, long M,
                                                    int L) {
  __shared__ float xs[(64 + 8) * DRK_V];
  __shared__ float pw[2][DRK_TAPS * DRK_V][DRK_C];   // 46 KB
  const long r0 = (long)blockIdx.x * 64;
  for (int i = threadIdx.x; i < (64 + 8) * DRK_V; i += 256) {
    const long r = r0 - 4 + i / DRK_V;
    const int v = i % DRK_V;
    float val = 0.f;
    if (r >= 0 && r < M) val = x ? ((int)x[r] == v ? 1.f : 0.f) : x_soft[r * DRK_V + v];
    xs[i] = val;
  }
  __syncthreads();
  const int co = threadIdx.x & 127, ph = threadIdx.x >> 7;
  float acc[DRK_TAPS * DRK_V];
#pragma unroll
  for (int k = 0; k < DRK_TAPS * DRK_V; ++k) acc[k] = 0.f;
  for (int rb = ph; rb < 64; rb += 16) {                 // 8 rows of dpre in flight per thread
    float dv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long row = r0 + rb + 2 * j;
      dv[j] = row < M ? dpre[row * DRK_C + co] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int rr = rb + 2 * j;
      const long row = r0 + rr;
      if (row >= M) break;
      const int p = (int)(row % L);
      const float d = dv[j];
#pragma unroll
      for (int t = 0; t < DRK_TAPS; ++t) {
        const int pp = p + t - 4;
        if (pp >= 0 && pp < L) {
#pragma unroll
          for (int v = 0; v < DRK_V; ++v) acc[t * DRK_V + v] += d * xs[(rr + t) * DRK_V + v];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < DRK_TAPS * DRK_V; ++k) pw[ph][k][co] = acc[k];
  __syncthreads();
  if (ph == 0) {
#pragma unroll
    for (int k = 0; k < DRK_TAPS * DRK_V; ++k) atomicAdd(dw_t + k * DRK_C + co, pw[0][k][co] + pw[1][k][co]);
  }
}

// k_stem_wgrad over SEVERAL recorded backward calls in one launch (the chain's stem gradient, formed once: drakes_amd/
// autograd.py::GradSink): workgroup = the same 64 rows of every segment, so the per-workgroup atomics are paid once per
// chain, not once per call; db[co] += the column sums of dpre as well (what k_colsum128 adds per call).
__global__ void __launch_bounds__(256) k_stem_wgrad_multi(const float* const* __restrict__ dpres, const int64_t* const* __restrict__ xs_tok,
                                                          const float* const* __restrict__ xs_soft, int n_seg,
                                                          float* __restrict__ dw_t, float* __restrict__ db, long M, int L) {
  __shared__ float xs[(64 + 8) * DRK_V];
  __shared__ float pw[2][DRK_TAPS * DRK_V + 1][DRK_C];
  const long r0 = (long)blockIdx.x * 64;
  const int co = threadIdx.x & 127, ph = threadIdx.x >> 7;
  float acc[DRK_TAPS * DRK_V];
#pragma unroll
  for (int k = 0; k < DRK_TAPS * DRK_V; ++k) acc[k] = 0.f;
  float bsum = 0.f;
  for (int seg = 0; seg < n_seg; ++seg) {
    const float* __restrict__ dpre = dpres[seg];
    const int64_t* __restrict__ x = xs_tok ? xs_tok[seg] : nullptr;
    const float* __restrict__ x_soft = xs_soft ? xs_soft[seg] : nullptr;
    __syncthreads();   // the previous segment's readers of xs are done
    for (int i = threadIdx.x; i < (64 + 8) * DRK_V; i += 256) {
      const long r = r0 - 4 + i / DRK_V;
      const int v = i % DRK_V;
      float val = 0.f;
      if (r >= 0 && r < M) val = x ? ((int)x[r] == v ? 1.f : 0.f) : x_soft[r * DRK_V + v];
      xs[i] = val;
    }
    __syncthreads();
    for (int rb = ph; rb < 64; rb += 16) {                 // 8 rows of dpre in flight per thread
      float dv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long row = r0 + rb + 2 * j;
        dv[j] = row < M ? dpre[row * DRK_C + co] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rr = rb + 2 * j;
        const long row = r0 + rr;
        if (row >= M) break;
        const int p = (int)(row % L);
        const float d = dv[j];
        bsum += d;
#pragma unroll
        for (int t = 0; t < DRK_TAPS; ++t) {
          const int pp = p + t - 4;
          if (pp >= 0 && pp < L) {
#pragma unroll
            for (int v = 0; v < DRK_V; ++v) acc[t * DRK_V + v] += d * xs[(rr + t) * DRK_V + v];
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < DRK_TAPS * DRK_V; ++k) pw[ph][k][co] = acc[k];
  pw[ph][DRK_TAPS * DRK_V][co] = bsum;
  __syncthreads();
  if (ph == 0) {
#pragma unroll
    for (int k = 0; k < DRK_TAPS * DRK_V; ++k) atomicAdd(dw_t + k * DRK_C + co, pw[0][k][co] + pw[1][k][co]);
    if (db) atomicAdd(db + co, pw[0][DRK_TAPS * DRK_V][co] + pw[1][DRK_TAPS * DRK_V][co]);
  }
}

// one wave per (row, all 5 v): lane = co pair; 9 taps; shuffle-reduce over the wave
__global__ void __launch_bounds__(256) k_stem_dgrad(const float* __restrict__ dpre, const float* __restrict__ w_t,
                                                    float* __restrict__ dx, long M, int L) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int p = (int)(row % L);
  float acc[DRK_V] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < DRK_TAPS; ++t) {
    const int pp = p - (t - 4);        // the output position this input fed through tap t
    if (pp >= 0 && pp < L) {
      const float* d = dpre + (row - (t - 4)) * DRK_C;
      const float d0 = d[lane], d1 = d[lane + 64];
#pragma unroll
      for (int v = 0; v < DRK_V; ++v)
        acc[v] += d0 * w_t[(t * DRK_V + v) * DRK_C + lane] + d1 * w_t[(t * DRK_V + v) * DRK_C + lane + 64];
    }
  }
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc[v] += __shfl_xor(acc[v], o);
  }
  if (lane < DRK_V) {
    float o = acc[0];
#pragma unroll
    for (int v = 1; v < DRK_V; ++v) o = (lane == v) ? acc[v] : o;
    dx[row * DRK_V + lane] = o;
  }
}

// ------------------------------------------------------------------------------------ time path backward
// tbias[b][l][c] = dense_w[l][c][:] . temb[b][:] + dense_b[l][c];  temb = relu(lin_w . four + lin_b); four = sin/cos(sigma*W*2pi)
// W (the Fourier frequencies) is not trained (dnaconv.py:17).  Two launches: one workgroup per (batch row, layer) for
// the Dense layers (their contributions to dtemb meet in `dtemb`, float32 [nb][128], zeroed by the caller), then one
// workgroup per batch row for the embedding's Linear.
__global__ void __launch_bounds__(128) k_time_bwd_dense(const float* __restrict__ dtbias, const float* __restrict__ temb,
                                                        const float* __restrict__ dense_w, float* __restrict__ d_dense_w,
                                                        float* __restrict__ d_dense_b, float* __restrict__ dtemb, int n_layers) {
  __shared__ float dt[DRK_C];
  const int b = blockIdx.x, l = blockIdx.y, c = threadIdx.x;
  const float te = temb[(long)b * DRK_C + c];
  dt[c] = dtbias[((long)b * n_layers + l) * DRK_C + c];
  __syncthreads();
  atomicAdd(d_dense_b + l * DRK_C + c, dt[c]);
  const float* wl = dense_w + (long)l * DRK_C * DRK_C;
  float* dwl = d_dense_w + (long)l * DRK_C * DRK_C;
  // thread c = INPUT index j here: dtemb[j] += sum_o dt[o] * w[o][j];  d_w[o][j] += dt[o] * te[j]
  float acc = 0.f;
  for (int o = 0; o < DRK_C; ++o) {
    acc += dt[o] * wl[o * DRK_C + c];
    atomicAdd(dwl + o * DRK_C + c, dt[o] * te);
  }
  atomicAdd(dtemb + (long)b * DRK_C + c, acc);
}

__global__ void __launch_bounds__(128) k_time_bwd_embed(const float* __restrict__ dtemb, const float* __restrict__ sigma,
                                                        const float* __restrict__ W, const float* __restrict__ temb,
                                                        float* __restrict__ d_lin_w, float* __restrict__ d_lin_b) {
  __shared__ float dpre[DRK_C];
  const int b = blockIdx.x, c = threadIdx.x;
  const float proj = ((sigma[b] * W[c & 63]) * 2.0f) * 3.14159274101257324f;
  const float four = (c < 64) ? sinf(proj) : cosf(proj);
  dpre[c] = temb[(long)b * DRK_C + c] > 0.f ? dtemb[(long)b * DRK_C + c] : 0.f;
  __syncthreads();
  atomicAdd(d_lin_b + c, dpre[c]);
  for (int o = 0; o < DRK_C; ++o) atomicAdd(d_lin_w + o * DRK_C + c, dpre[o] * four);
}

// ------------------------------------------------------------------------------------ SUBS backward
// lp = SUBS(logits, tok): for masked rows dlogits_i = dlp_i - p_i * sum_{j<4} dlp_j (i<4, p = exp(lp)), dlogits_4 = 0
// (its probability is exp(-1e6) = 0); rows holding a token are constants: dlogits = 0  (diffusion_gosai_update.py:250-252).
__device__ __forceinline__ void subs_bwd_row(const float lp[DRK_V], const float dlp[DRK_V], int tok, float dl[DRK_V]) {
  if (tok != DRK_MASK) {
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) dl[v] = 0.f;
    return;
  }
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) s += dlp[v];
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) dl[v] = dlp[v] - expf(lp[v]) * s;
}

__device__ __forceinline__ int tok_of(const int64_t* x, const float* x_soft, long i) {
  if (x) return (int)x[i];
  int best = 0;
#pragma unroll
  for (int k = 1; k < DRK_V; ++k)
    if (x_soft[i * DRK_V + k] > x_soft[i * DRK_V + best]) best = k;
  return best;
}

__global__ void __launch_bounds__(256) k_subs_bwd(const float* __restrict__ logits, const int64_t* __restrict__ x,
                                                  const float* __restrict__ x_soft, const float* __restrict__ dlogp,
                                                  float* __restrict__ dlogits, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float lg[DRK_V], lp[DRK_V], dlp[DRK_V], dl[DRK_V];
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) {
      lg[v] = logits[i * DRK_V + v];
      dlp[v] = dlogp[i * DRK_V + v];
    }
    const int tok = tok_of(x, x_soft, i);
    drk_subs_logp(lg, tok, lp);
    subs_bwd_row(lp, dlp, tok, dl);
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) dlogits[i * DRK_V + v] = dl[v];
  }
}

// ------------------------------------------------------------------------------------ soft reverse step
// _ddpm_update_finetune_gradient tail (diffusion_gosai_update.py:572-586) + _sample_categorical_gradient (:30-35):
//   lp = SUBS(logits, argmax x);  q = exp(lp) * (mc_t - mc_s), q[mask] = mc_s;  g = 1e-10 - log(U + 1e-10)
//   y = softmax((log q - log g) / temp);  flag = 1 - x[mask]  (or sigmoid((1 - x[mask]) / cft))
//   x_next = flag * x + (1 - flag) * y
__device__ __forceinline__ void soft_row(const float lg[DRK_V], const float xs[DRK_V], const float u[DRK_V], float mc_t,
                                         float mc_s, float inv_temp, float cft, int& tok, float lp[DRK_V], float y[DRK_V],
                                         float& flag) {
  tok = 0;
#pragma unroll
  for (int k = 1; k < DRK_V; ++k)
    if (xs[k] > xs[tok]) tok = k;
  drk_subs_logp(lg, tok, lp);
  float q[DRK_V], z[DRK_V];
  drk_posterior(lp, mc_t, mc_s, q);
  float m = -INFINITY;
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) {
    const float g = 1e-10f - logf(u[v] + 1e-10f);
    z[v] = (logf(q[v]) - logf(g)) * inv_temp;
    m = fmaxf(m, z[v]);
  }
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) {
    y[v] = expf(z[v] - m);
    s += y[v];
  }
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) y[v] = y[v] / s;
  const float cp = 1.0f - xs[DRK_MASK];
  flag = (cft > 0.f) ? 1.0f / (1.0f + expf(-cp / cft)) : cp;
}

template <bool INJECT>
__global__ void __launch_bounds__(256)
    k_soft_step_fwd(const float* __restrict__ logits, const float* __restrict__ x_soft, const float* __restrict__ uniforms,
                    float* __restrict__ x_next, float* __restrict__ flag_out, long n, int L, float mc_t, float mc_s,
                    const float* __restrict__ mc_rows, float inv_temp, float cft, uint64_t seed, uint32_t step,
                    uint64_t sample_offset) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float lg[DRK_V], xs[DRK_V], u[DRK_V], lp[DRK_V], y[DRK_V], flag;
    int tok;
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) {
      lg[v] = logits[i * DRK_V + v];
      xs[v] = x_soft[i * DRK_V + v];
    }
    const uint32_t row = (uint32_t)i / (uint32_t)L;   // B*L < 2^32, checked by the entry point
    if (INJECT) {
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) u[v] = uniforms[i * DRK_V + v];
    } else {
      drk_step_uniforms(seed, step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, u);
    }
    if (mc_rows) {   // per-row timesteps
      mc_t = mc_rows[2 * row];
      mc_s = mc_rows[2 * row + 1];
    }
    soft_row(lg, xs, u, mc_t, mc_s, inv_temp, cft, tok, lp, y, flag);
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) x_next[i * DRK_V + v] = flag * xs[v] + (1.0f - flag) * y[v];
    flag_out[i] = flag;
  }
}

// Backward: given G = dL/dx_next (and optionally Gf = dL/dflag from a use of the returned flag):
//   dL/dx (direct part) and dL/dlogits; the forward row is recomputed from the same inputs.
template <bool INJECT>
__global__ void __launch_bounds__(256)
    k_soft_step_bwd(const float* __restrict__ logits, const float* __restrict__ x_soft, const float* __restrict__ uniforms,
                    const float* __restrict__ gnext, const float* __restrict__ gflag, float* __restrict__ dlogits,
                    float* __restrict__ dx, long n, int L, float mc_t, float mc_s, const float* __restrict__ mc_rows,
                    float inv_temp, float cft, uint64_t seed, uint32_t step, uint64_t sample_offset) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float lg[DRK_V], xs[DRK_V], u[DRK_V], lp[DRK_V], y[DRK_V], G[DRK_V], flag;
    int tok;
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) {
      lg[v] = logits[i * DRK_V + v];
      xs[v] = x_soft[i * DRK_V + v];
      G[v] = gnext[i * DRK_V + v];
    }
    const uint32_t row = (uint32_t)i / (uint32_t)L;   // B*L < 2^32, checked by the entry point
    if (INJECT) {
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) u[v] = uniforms[i * DRK_V + v];
    } else {
      drk_step_uniforms(seed, step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, u);
    }
    if (mc_rows) {   // per-row timesteps
      mc_t = mc_rows[2 * row];
      mc_s = mc_rows[2 * row + 1];
    }
    soft_row(lg, xs, u, mc_t, mc_s, inv_temp, cft, tok, lp, y, flag);
    float dflag = gflag ? gflag[i] : 0.f;
    float dyy = 0.f, dy[DRK_V], dxo[DRK_V];
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) {
      dflag += G[v] * (xs[v] - y[v]);
      dy[v] = (1.0f - flag) * G[v];
      dyy += y[v] * dy[v];
      dxo[v] = flag * G[v];
    }
    dxo[DRK_MASK] += (cft > 0.f) ? -dflag * flag * (1.0f - flag) / cft : -dflag;
    float dlp[DRK_V], dl[DRK_V];
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) dlp[v] = y[v] * (dy[v] - dyy) * inv_temp;   // d/d log q = d/d lp for v < 4
    dlp[DRK_MASK] = 0.f;                                                        // q[mask] = mc_s is a constant
    subs_bwd_row(lp, dlp, tok, dl);
#pragma unroll
    for (int v = 0; v < DRK_V; ++v) {
      dlogits[i * DRK_V + v] = dl[v];
      dx[i * DRK_V + v] = dxo[v];
    }
  }
}

// ---- per-step KL term of the fine-tune loss (finetune_reward_bp.py:67-74) ---------------------------
//   kl[b] = sum_{l, v<4} flag[b,l] * (-p + p_old + p * (lp - lp_old)) / mc_t * last_x[b,l,v],  p = exp(lp)
// One workgroup per batch row (deterministic reduction order).  flag is int64 (hard steps) or float (soft steps).
__device__ __forceinline__ float kl_flag(const float* ff, const int64_t* fi, long i) { return ff ? ff[i] : (float)fi[i]; }

__global__ void __launch_bounds__(256) k_kl_fwd(const float* __restrict__ lp, const float* __restrict__ lpo,
                                                const float* __restrict__ x, const float* __restrict__ ff,
                                                const int64_t* __restrict__ fi, float inv_mc, float* __restrict__ kl, int L) {
  const long b = blockIdx.x;
  float acc = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    const long i = b * L + l;
    const float cf = kl_flag(ff, fi, i);
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < DRK_V - 1; ++v) {
      const float a = lp[i * DRK_V + v], o = lpo[i * DRK_V + v];
      const float p = expf(a), po = expf(o);
      s += (cf * (-p + po + p * (a - o)) * inv_mc) * x[i * DRK_V + v];
    }
    acc += s;
  }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) kl[b] = red[0];
}

// d kl[b] / d lp, d lp_old, d last_x, d flag, scaled by g[b]; dlpo / dx / dflag may be NULL (hard steps: constants).
__global__ void __launch_bounds__(256) k_kl_bwd(const float* __restrict__ lp, const float* __restrict__ lpo,
                                                const float* __restrict__ x, const float* __restrict__ ff,
                                                const int64_t* __restrict__ fi, const float* __restrict__ g, float inv_mc,
                                                float* __restrict__ dlp, float* __restrict__ dlpo, float* __restrict__ dx,
                                                float* __restrict__ dflag, long n, int L) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gb = g[i / L] * inv_mc;
    const float cf = kl_flag(ff, fi, i);
    float df = 0.f;
#pragma unroll
    for (int v = 0; v < DRK_V - 1; ++v) {
      const float a = lp[i * DRK_V + v], o = lpo[i * DRK_V + v], xv = x[i * DRK_V + v];
      const float p = expf(a), po = expf(o);
      const float term = -p + po + p * (a - o);
      dlp[i * DRK_V + v] = gb * cf * xv * (p * (a - o));
      if (dlpo) dlpo[i * DRK_V + v] = gb * cf * xv * (po - p);
      if (dx) dx[i * DRK_V + v] = gb * cf * term;
      df += term * xv;
    }
    dlp[i * DRK_V + DRK_MASK] = 0.f;
    if (dlpo) dlpo[i * DRK_V + DRK_MASK] = 0.f;
    if (dx) dx[i * DRK_V + DRK_MASK] = 0.f;
    if (dflag) dflag[i] = gb * df;
  }
}

// ---- wide vocabulary (protein, V = 22): the Gumbel-softmax step of Interpolant.sample_gradient ----------------
// drakes_protein/fmif/fm_utils.py:159-192 (soft branch) + _sample_categorical_gradient (:21-26):
//   l = logits / temp with the mask logit at -1e6;  lp = l - logsumexp(l), or (-1e6 .. 0 at argmax x .. -1e6) when
//   argmax x is not the mask;  q = exp(lp) * d_t, q[mask] = move_chance_s, q += 1e-8;  g = 1e-10 - log(U + 1e-10);
//   y = softmax((log q - log g) / gumbel_temp);  flag = 1 - x[mask];  x_next = x * flag + y * (1 - flag)
template <int V>
__device__ __forceinline__ void wide_uniforms(uint64_t seed, uint32_t step, uint64_t sample, uint32_t pos, float (&u)[V]) {
  const float sc = 5.9604644775390625e-08f;
#pragma unroll
  for (int blk = 0; 4 * blk < V; ++blk) {
    uint32_t r[4];
    drk_philox4x32_10(pos, (uint32_t)sample, step, 0x50524F54u ^ (uint32_t)(sample >> 32) ^ ((uint32_t)blk << 24),
                      (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (4 * blk + w < V) u[4 * blk + w] = (float)(r[w] >> 8) * sc;
  }
}

template <int V>
__device__ __forceinline__ void soft_row_wide(const float (&lg)[V], const float (&xs)[V], const float (&u)[V], int mask,
                                              float temp, float d_t, float mc_s, float inv_gtemp, int& tok, float (&lp)[V],
                                              float (&q)[V], float (&y)[V], float& flag) {
  tok = 0;
#pragma unroll
  for (int k = 1; k < V; ++k)
    if (xs[k] > xs[tok]) tok = k;
  float l[V], m = -INFINITY;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    l[v] = ((v == mask) ? DRK_NEG_INF : lg[v]) / temp;
    m = fmaxf(m, l[v]);
  }
  const float ms = (fabsf(m) == INFINITY) ? 0.0f : m;
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < V; ++v) s += expf(l[v] - ms);
  const float lse = logf(s) + ms;
  float z[V], zm = -INFINITY;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    lp[v] = (tok != mask) ? ((v == tok) ? 0.0f : DRK_NEG_INF) : l[v] - lse;
    q[v] = ((v == mask) ? mc_s : expf(lp[v]) * d_t) + 1e-8f;
    const float g = 1e-10f - logf(u[v] + 1e-10f);
    z[v] = (logf(q[v]) - logf(g)) * inv_gtemp;
    zm = fmaxf(zm, z[v]);
  }
  float ys = 0.f;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    y[v] = expf(z[v] - zm);
    ys += y[v];
  }
#pragma unroll
  for (int v = 0; v < V; ++v) y[v] = y[v] / ys;
  flag = 1.0f - xs[mask];
}

template <int V, bool INJECT>
__global__ void __launch_bounds__(128)
    k_soft_step_wide_fwd(const float* __restrict__ logits, const float* __restrict__ x_soft, const float* __restrict__ uniforms,
                         float* __restrict__ x_next, float* __restrict__ flag_out, long n, int L, int mask, float temp,
                         float d_t, float mc_s, float inv_gtemp, uint64_t seed, uint32_t step, uint64_t sample_offset) {
  for (long i = (long)blockIdx.x * 128 + threadIdx.x; i < n; i += (long)gridDim.x * 128) {
    float lg[V], xs[V], u[V], lp[V], q[V], y[V], flag;
    int tok;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      lg[v] = logits[i * V + v];
      xs[v] = x_soft[i * V + v];
    }
    if (INJECT) {
#pragma unroll
      for (int v = 0; v < V; ++v) u[v] = uniforms[i * V + v];
    } else {
      const uint32_t row = (uint32_t)i / (uint32_t)L;
      wide_uniforms<V>(seed, step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, u);
    }
    soft_row_wide<V>(lg, xs, u, mask, temp, d_t, mc_s, inv_gtemp, tok, lp, q, y, flag);
#pragma unroll
    for (int v = 0; v < V; ++v) x_next[i * V + v] = xs[v] * flag + y[v] * (1.0f - flag);
    flag_out[i] = flag;
  }
}

template <int V, bool INJECT>
__global__ void __launch_bounds__(128)
    k_soft_step_wide_bwd(const float* __restrict__ logits, const float* __restrict__ x_soft, const float* __restrict__ uniforms,
                         const float* __restrict__ gnext, const float* __restrict__ gflag, float* __restrict__ dlogits,
                         float* __restrict__ dx, long n, int L, int mask, float temp, float d_t, float mc_s, float inv_gtemp,
                         uint64_t seed, uint32_t step, uint64_t sample_offset) {
  for (long i = (long)blockIdx.x * 128 + threadIdx.x; i < n; i += (long)gridDim.x * 128) {
    float lg[V], xs[V], u[V], lp[V], q[V], y[V], G[V], flag;
    int tok;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      lg[v] = logits[i * V + v];
      xs[v] = x_soft[i * V + v];
      G[v] = gnext[i * V + v];
    }
    if (INJECT) {
#pragma unroll
      for (int v = 0; v < V; ++v) u[v] = uniforms[i * V + v];
    } else {
      const uint32_t row = (uint32_t)i / (uint32_t)L;
      wide_uniforms<V>(seed, step, sample_offset + row, (uint32_t)i - row * (uint32_t)L, u);
    }
    soft_row_wide<V>(lg, xs, u, mask, temp, d_t, mc_s, inv_gtemp, tok, lp, q, y, flag);
    float dflag = gflag ? gflag[i] : 0.f, dyy = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      dflag += G[v] * (xs[v] - y[v]);
      dyy += y[v] * ((1.0f - flag) * G[v]);
    }
    float dlp[V], sdl = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const float dz = y[v] * ((1.0f - flag) * G[v] - dyy);
      const float e = expf(lp[v]) * d_t;                    // q = e + 1e-8: d log q / d lp = e / q
      dlp[v] = (v == mask || tok != mask) ? 0.f : dz * inv_gtemp * (e / q[v]);
      sdl += dlp[v];
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      dlogits[i * V + v] = (v == mask || tok != mask) ? 0.f : (dlp[v] - expf(lp[v]) * sdl) / temp;
      dx[i * V + v] = flag * G[v] - ((v == mask) ? dflag : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" int drk_mask_mul(const float* g, const unsigned char* mask, float* out, long n, void* stream) {
  ARGCHECK(n >= 0 && (n % 4) == 0, "drk_mask_mul");
  if (n == 0) return DRK_OK;
  ARGCHECK(g && mask && out, "drk_mask_mul");
  hipLaunchKernelGGL(k_mask_mul, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, g, mask, out, n / 4);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_relu_bwd(const float* g, const float* y, float* out, long n, void* stream) {
  ARGCHECK(n >= 0 && (n % 4) == 0, "drk_relu_bwd");
  if (n == 0) return DRK_OK;
  ARGCHECK(g && y && out, "drk_relu_bwd");
  hipLaunchKernelGGL(k_relu_bwd, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, g, y, out, n / 4);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_colsum128(const float* x, float* out, long n_seg, long seg_rows, void* stream) {
  ARGCHECK(n_seg >= 0 && seg_rows >= 0 && n_seg < 65536, "drk_colsum128");
  if (n_seg == 0 || seg_rows == 0) return DRK_OK;
  ARGCHECK(x && out, "drk_colsum128");
  hipLaunchKernelGGL(k_colsum128, dim3((unsigned)((seg_rows + CS_ROWS - 1) / CS_ROWS), (unsigned)n_seg), dim3(256), 0,
                     (hipStream_t)stream, x, out, seg_rows);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_ln_bwd(const float* dh, const float* feat, const float* tbias, long tbias_bstride, const float* gamma,
                          const float* mean, const float* rstd, const float* g_resid, float* dxin, float* dfeat,
                          float* dgamma, float* dbeta, float* dtbias_shared, const unsigned char* next_mask, void* dc_next,
                          float* dbias_next, int B, int L, int width, void* stream) {
  DRK_SHAPE_GUARD("drk_ln_bwd");
  ARGCHECK(width >= 4 && width <= DRK_C && width % 4 == 0, "drk_ln_bwd (width: a multiple of 4 in [4, 128])");
  ARGCHECK(dh && feat && gamma && mean && rstd && dfeat && dgamma && dbeta, "drk_ln_bwd");
  ARGCHECK((!next_mask && !dc_next && !dbias_next) || (next_mask && dc_next), "drk_ln_bwd (next_mask and dc_next come together)");
  const long M = (long)B * L;
#ifndef LNB_WGS
#define LNB_WGS 256L
#endif
  long passes = (M + LNB_WGS * LNB_PASS_ROWS - 1) / (LNB_WGS * LNB_PASS_ROWS);   // about one workgroup per CU of an MI355X
  passes = passes < 1 ? 1 : passes > 8 ? 8 : passes;
  const long wg_rows = passes * LNB_PASS_ROWS;
  hipLaunchKernelGGL(k_ln_bwd, dim3((unsigned)((M + wg_rows - 1) / wg_rows)), dim3(512), 0, (hipStream_t)stream, dh, feat, tbias,
                     tbias_bstride, gamma, mean, rstd, g_resid, dxin, dfeat, dgamma, dbeta, dtbias_shared, next_mask,
                     (char*)dc_next, dbias_next, M, L, width, (int)passes);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

int drk_slab_reduce_launch(const float* slabs, int n_slabs, long stride, float* out, long n, hipStream_t stream) {
  const long n4 = n / 4;
  long grid = (n4 + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)grid), dim3(256), 0, stream, slabs, n_slabs, stride, out, n4);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

// Stage 1 of the two-stage form: group g = blockIdx.y sums slabs [g * per, min((g + 1) * per, n_slabs)) in order and STORES
// the sum as slab g of `tmp` (stride n): many slabs of a short vector then run on groups x columns workgroups instead of
// a dozen, each with a chain of n_slabs dependent-latency loads.  Stage 2 is k_slab_reduce over the groups.
__global__ void __launch_bounds__(256) k_slab_reduce_groups(const float* __restrict__ slabs, int n_slabs, long stride, int per,
                                                            float* __restrict__ tmp, long n4) {
  const int g = blockIdx.y;
  const int s0 = g * per, s1 = (s0 + per < n_slabs) ? s0 + per : n_slabs;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = s0; k0 < s1; k0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k0 + j < s1) v[j] = ((const f32x4*)(slabs + (long)(k0 + j) * stride))[i];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    ((f32x4*)(tmp + (long)g * n4 * 4))[i] = s;
  }
}

// out[i] += sum over slabs, in a FIXED order (groups of `DRK_REDUCE_GROUP_SLABS` consecutive slabs, then the groups):
// tmp holds drk_slab_reduce2_tmp_floats(n_slabs, n) floats.
#define DRK_REDUCE_GROUP_SLABS 16
extern "C" long drk_slab_reduce2_tmp_floats(int n_slabs, long n) {
  if (n_slabs <= 0 || n <= 0) return 0;
  return (long)((n_slabs + DRK_REDUCE_GROUP_SLABS - 1) / DRK_REDUCE_GROUP_SLABS) * n;
}

extern "C" int drk_slab_reduce2(const float* slabs, int n_slabs, long stride, float* out, long n, float* tmp, void* stream) {
  if (n_slabs < 0 || n < 0 || (n & 3) || (n && (!slabs || !out || !tmp)) || (((uintptr_t)slabs | (uintptr_t)out | (uintptr_t)tmp) & 15) ||
      (stride & 3))
    return drk_set_error_msg(DRK_EINVAL, "drk_slab_reduce2: n % 4 == 0, stride % 4 == 0, 16-byte aligned pointers, tmp given");
  if (n == 0 || n_slabs == 0) return DRK_OK;
  const int groups = (n_slabs + DRK_REDUCE_GROUP_SLABS - 1) / DRK_REDUCE_GROUP_SLABS;
  const long n4 = n / 4;
  long gx = (n4 + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(k_slab_reduce_groups, dim3((unsigned)gx, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs,
                     stride, DRK_REDUCE_GROUP_SLABS, tmp, n4);
  DRK_CHECK_LAUNCH();
  return drk_slab_reduce_launch(tmp, groups, n, out, n, (hipStream_t)stream);
}

extern "C" int drk_slab_reduce(const float* slabs, int n_slabs, long stride, float* out, long n, void* stream) {
  if (n_slabs < 0 || n < 0 || (n & 3) || (n && (!slabs || !out)) || (((uintptr_t)slabs | (uintptr_t)out) & 15) || (stride & 3))
    return drk_set_error_msg(DRK_EINVAL, "drk_slab_reduce: n % 4 == 0, stride % 4 == 0, 16-byte aligned pointers");
  if (n == 0 || n_slabs == 0) return DRK_OK;
  return drk_slab_reduce_launch(slabs, n_slabs, stride, out, n, (hipStream_t)stream);
}

// Row chunk of the split-K weight gradient: the smallest 64 * 2^k rows with at most WG_MAX_SLABS (chunk, tap)
// workgroups — two co-resident workgroups per CU (the loop is load-latency-bound: a second workgroup hides it; one
// round of 1024-row workgroups measured 27 % slower at B = 128), and a bounded workspace (one 64-KB slab each).
#define WG_MAX_SLABS 504
#define WG_MAX_TAPS 9
static int wgrad_rows_per_chunk(long M, int taps) {
  long wg_rows = 64;
  while (((M + wg_rows - 1) / wg_rows) * taps > WG_MAX_SLABS) wg_rows <<= 1;
  return (int)wg_rows;
}
static long wgrad_ws_floats(long M, int taps) {
  const long chunks = (M + wgrad_rows_per_chunk(M, taps) - 1) / wgrad_rows_per_chunk(M, taps);
  return chunks * ((long)taps * WG_SLAB + DRK_C);
}

extern "C" long drk_conv_wgrad_workspace_bytes(int B, int L, int taps) {
  if (B <= 0 || L <= 0 || taps < 1 || taps > WG_MAX_TAPS) return 0;
  return wgrad_ws_floats((long)B * L, taps) * (long)sizeof(float);
}

// Shared by the workspace form and the allocation-free form: slabs in `ws` (chunks x taps x 128 x 128 floats, then
// chunks x 128 bias partials), then the ordered reduction into dwt / dbias.
static int wgrad_launch(const float* h, const float* dc, float* dwt, float* dbias, long M, int L, int taps, int dilation,
                        float* ws, hipStream_t stream) {
  const int wg_rows = wgrad_rows_per_chunk(M, taps);
  const long chunks = (M + wg_rows - 1) / wg_rows;
  float* bslabs = ws + chunks * taps * WG_SLAB;
  hipLaunchKernelGGL(k_conv_wgrad, dim3((unsigned)chunks, (unsigned)taps), dim3(256), 0, stream, h, dc, ws,
                     dbias ? bslabs : nullptr, M, L, taps, dilation, wg_rows);
  DRK_CHECK_LAUNCH();
  int rc = drk_slab_reduce_launch(ws, (int)chunks, (long)taps * WG_SLAB, dwt, (long)taps * WG_SLAB, stream);
  if (rc == DRK_OK && dbias) rc = drk_slab_reduce_launch(bslabs, (int)chunks, (long)DRK_C, dbias, (long)DRK_C, stream);
  return rc;
}

extern "C" int drk_conv_wgrad_f32(const float* h, const float* dc, float* dwt, float* dbias, int B, int L, int taps,
                                  int dilation, void* workspace, long workspace_bytes, void* stream) {
  DRK_SHAPE_GUARD("drk_conv_wgrad_f32");
  ARGCHECK(h && dc && dwt && taps >= 1 && taps <= WG_MAX_TAPS && (taps & 1) && dilation >= 0, "drk_conv_wgrad_f32");
  ARGCHECK(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= drk_conv_wgrad_workspace_bytes(B, L, taps),
           "drk_conv_wgrad_f32 (workspace: 16-byte aligned, drk_conv_wgrad_workspace_bytes(B, L, taps) bytes)");
  return wgrad_launch(h, dc, dwt, dbias, (long)B * L, L, taps, dilation, (float*)workspace, (hipStream_t)stream);
}

// ---- the same, allocating nothing and taking no workspace -------------------------------------------------------
// For callers that cannot call hipMalloc or carry a workspace (code under hipGraph capture, tight serving loops,
// harnesses that forbid allocation).  The partial slabs live in a file-scope device array: WG_MAX_SLABS slabs of
// 64 KB + their bias partials = 33.3 MB, held by every GPU the library runs on until the process ends.  In host code
// the array's NAME is only a handle: its device address is taken once per device with hipGetSymbolAddress (each GPU
// holds its own copy at its own address) and only that pointer is ever handed to a launch.
#define WG_SCRATCH_FLOATS ((long)WG_MAX_SLABS * WG_SLAB + (long)WG_MAX_SLABS * DRK_C)
#define WG_STATIC_MAX_ROWS (1L << 22)   // B * L accepted by the allocation-free form
static __device__ __align__(256) float scratch[WG_SCRATCH_FLOATS];
static std::atomic<float*> g_scratch_ptr[DRK_MAX_DEVICES];   // per device, set on the first call with that device current

extern "C" int drk_conv_wgrad_f32_static(const float* h, const float* dc, float* dwt, float* dbias, int B, int L, int taps,
                                         int dilation, void* stream) {
  DRK_SHAPE_GUARD("drk_conv_wgrad_f32_static");
  ARGCHECK(h && dc && dwt && taps >= 1 && (taps & 1) && dilation >= 0, "drk_conv_wgrad_f32_static");
  const long M = (long)B * L;
  if (taps > WG_MAX_TAPS || M > WG_STATIC_MAX_ROWS)
    return drk_set_error_msg(DRK_EINVAL, "drk_conv_wgrad_f32_static: beyond the built-in scratch (taps <= 9, B*L <= 4194304 rows); "
                                         "use drk_conv_wgrad_f32 with a workspace");
  if (wgrad_ws_floats(M, taps) > WG_SCRATCH_FLOATS)
    return drk_set_error_msg(DRK_EINVAL, "drk_conv_wgrad_f32_static: scratch too small for this shape");
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return drk_set_error(e, "drk_conv_wgrad_f32_static(hipGetDevice)");
  if (dev < 0 || dev >= DRK_MAX_DEVICES) return drk_set_error_msg(DRK_EINVAL, "drk_conv_wgrad_f32_static: device index beyond DRK_MAX_DEVICES");
  float* ws = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!ws) {   // first call on this device: resolve this device's copy of the array (not an allocation)
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) return drk_set_error(e, "drk_conv_wgrad_f32_static(hipGetSymbolAddress)");
    ws = (float*)p;
    g_scratch_ptr[dev].store(ws, std::memory_order_release);
  }
  // every slab the reduction reads is fully written by this call's first launch, so the previous call's contents
  // need no clearing; two calls overlapping on one GPU (different streams) would race on the array.
  return wgrad_launch(h, dc, dwt, dbias, M, L, taps, dilation, ws, (hipStream_t)stream);
}

extern "C" int drk_head_out_bwd(const float* dlogits, const float* y, const float* w, float* dy, float* dw, float* db,
                                int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_head_out_bwd");
  ARGCHECK(dlogits && y && w && dy && dw && db, "drk_head_out_bwd");
  const long M = (long)B * L;
  hipLaunchKernelGGL(k_head_out_bwd, dim3((unsigned)((M + CS_ROWS - 1) / CS_ROWS)), dim3(256), 0, (hipStream_t)stream, dlogits, y, w,
                     dy, dw, db, M);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_stem_bwd(const float* dpre, const int64_t* x, const float* x_soft, const float* w_t, float* dw_t,
                            float* dx, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_stem_bwd");
  ARGCHECK(dpre && (x || x_soft) && w_t && (dw_t || dx), "drk_stem_bwd");
  const long M = (long)B * L;
  if (dw_t) {   // (NULL: a frozen model passes only the input gradient on)
    hipLaunchKernelGGL(k_stem_wgrad, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dpre, x, x_soft,
                       dw_t, M, L);
    DRK_CHECK_LAUNCH();
  }
  if (dx) {
    hipLaunchKernelGGL(k_stem_dgrad, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dpre, w_t, dx, M, L);
    DRK_CHECK_LAUNCH();
  }
  return DRK_OK;
}

// The stem's weight and bias gradients of n_seg recorded backward calls of one shape in one launch: dw_t [9][5][128] +=,
// db [128] += (NULL: not wanted); dpre_ptrs / x_ptrs / x_soft_ptrs: DEVICE arrays of n_seg pointers to each call's dpre
// [B*L][128] and its input (tokens int64 [B*L] through x_ptrs, or soft float32 [B*L][5] through x_soft_ptrs: one of the
// two arrays, the other NULL).
extern "C" int drk_stem_wgrad_multi(const float* const* dpre_ptrs, const int64_t* const* x_ptrs, const float* const* x_soft_ptrs,
                                    int n_seg, float* dw_t, float* db, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_stem_wgrad_multi");
  if (n_seg == 0) return DRK_OK;
  ARGCHECK(n_seg > 0 && dpre_ptrs && (!x_ptrs != !x_soft_ptrs) && dw_t, "drk_stem_wgrad_multi");
  const long M = (long)B * L;
  hipLaunchKernelGGL(k_stem_wgrad_multi, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dpre_ptrs, x_ptrs,
                     x_soft_ptrs, n_seg, dw_t, db, M, L);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_time_bwd(const float* dtbias, const float* sigma, int nb, const float* W, const float* temb,
                            const float* dense_w, int n_layers, float* d_dense_w, float* d_dense_b, float* d_lin_w,
                            float* d_lin_b, float* dtemb, void* stream) {
  ARGCHECK(nb >= 0 && n_layers > 0 && n_layers < 65536, "drk_time_bwd");
  if (nb == 0) return DRK_OK;
  ARGCHECK(dtbias && sigma && W && temb && dense_w && d_dense_w && d_dense_b && d_lin_w && d_lin_b && dtemb, "drk_time_bwd");
  hipLaunchKernelGGL(k_time_bwd_dense, dim3(nb, n_layers), dim3(128), 0, (hipStream_t)stream, dtbias, temb, dense_w,
                     d_dense_w, d_dense_b, dtemb, n_layers);
  DRK_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_time_bwd_embed, dim3(nb), dim3(128), 0, (hipStream_t)stream, dtemb, sigma, W, temb, d_lin_w, d_lin_b);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_subs_bwd(const float* logits, const int64_t* x, const float* x_soft, const float* dlogp,
                            float* dlogits, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_subs_bwd");
  ARGCHECK(logits && (x || x_soft) && dlogp && dlogits, "drk_subs_bwd");
  const long n = (long)B * L;
  hipLaunchKernelGGL(k_subs_bwd, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, x, x_soft, dlogp,
                     dlogits, n);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_reverse_step_soft_fwd(const float* logits, const float* x_soft, const float* uniforms, float* x_next,
                                         float* flag, int B, int L, float mc_t, float mc_s, const float* mc_rows,
                                         float temp, float copy_flag_temp, uint64_t seed, uint32_t step,
                                         uint64_t sample_offset, void* stream) {
  DRK_SHAPE_GUARD("drk_reverse_step_soft_fwd");
  ARGCHECK(logits && x_soft && x_next && flag && temp > 0.f, "drk_reverse_step_soft_fwd");
  const long n = (long)B * L;
  ARGCHECK(n < (1L << 32), "drk_reverse_step_soft_fwd (B*L must be below 2^32)");
  if (uniforms)
    hipLaunchKernelGGL(k_soft_step_fwd<true>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, x_soft,
                       uniforms, x_next, flag, n, L, mc_t, mc_s, mc_rows, 1.0f / temp, copy_flag_temp, seed, step,
                       sample_offset);
  else
    hipLaunchKernelGGL(k_soft_step_fwd<false>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, x_soft,
                       uniforms, x_next, flag, n, L, mc_t, mc_s, mc_rows, 1.0f / temp, copy_flag_temp, seed, step,
                       sample_offset);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_reverse_step_soft_bwd(const float* logits, const float* x_soft, const float* uniforms,
                                         const float* grad_x_next, const float* grad_flag, float* dlogits, float* dx, int B,
                                         int L, float mc_t, float mc_s, const float* mc_rows, float temp,
                                         float copy_flag_temp, uint64_t seed, uint32_t step, uint64_t sample_offset,
                                         void* stream) {
  DRK_SHAPE_GUARD("drk_reverse_step_soft_bwd");
  ARGCHECK(logits && x_soft && grad_x_next && dlogits && dx && temp > 0.f, "drk_reverse_step_soft_bwd");
  const long n = (long)B * L;
  ARGCHECK(n < (1L << 32), "drk_reverse_step_soft_bwd (B*L must be below 2^32)");
  if (uniforms)
    hipLaunchKernelGGL(k_soft_step_bwd<true>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, x_soft,
                       uniforms, grad_x_next, grad_flag, dlogits, dx, n, L, mc_t, mc_s, mc_rows, 1.0f / temp, copy_flag_temp,
                       seed, step, sample_offset);
  else
    hipLaunchKernelGGL(k_soft_step_bwd<false>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, logits, x_soft,
                       uniforms, grad_x_next, grad_flag, dlogits, dx, n, L, mc_t, mc_s, mc_rows, 1.0f / temp, copy_flag_temp,
                       seed, step, sample_offset);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_kl_step_fwd(const float* logp, const float* logp_old, const float* last_x, const float* flag_f,
                               const int64_t* flag_i, float move_chance_t, float* kl, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_kl_step_fwd");
  ARGCHECK(logp && logp_old && last_x && (flag_f || flag_i) && kl && move_chance_t > 0.f, "drk_kl_step_fwd");
  hipLaunchKernelGGL(k_kl_fwd, dim3(B), dim3(256), 0, (hipStream_t)stream, logp, logp_old, last_x, flag_f, flag_i,
                     1.0f / move_chance_t, kl, L);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_kl_step_bwd(const float* logp, const float* logp_old, const float* last_x, const float* flag_f,
                               const int64_t* flag_i, const float* grad_kl, float move_chance_t, float* dlogp,
                               float* dlogp_old, float* dlast_x, float* dflag, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_kl_step_bwd");
  ARGCHECK(logp && logp_old && last_x && (flag_f || flag_i) && grad_kl && dlogp && move_chance_t > 0.f, "drk_kl_step_bwd");
  const long n = (long)B * L;
  hipLaunchKernelGGL(k_kl_bwd, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, logp, logp_old, last_x, flag_f,
                     flag_i, grad_kl, 1.0f / move_chance_t, dlogp, dlogp_old, dlast_x, dflag, n, L);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_reverse_step_soft_wide_fwd(const float* logits, const float* x_soft, const float* uniforms, float* x_next,
                                              float* flag, int B, int L, int V, int mask_index, float temp, float d_t,
                                              float move_chance_s, float gumbel_temp, uint64_t seed, uint32_t step,
                                              uint64_t sample_offset, void* stream) {
  DRK_SHAPE_GUARD("drk_reverse_step_soft_wide_fwd");
  ARGCHECK(logits && x_soft && x_next && flag && temp > 0.f && gumbel_temp > 0.f && V == 22 && mask_index >= 0 &&
               mask_index < V, "drk_reverse_step_soft_wide_fwd (V must be 22)");
  const long n = (long)B * L;
  ARGCHECK(n < (1L << 32), "drk_reverse_step_soft_wide_fwd (B*L must be below 2^32)");
  if (uniforms)
    hipLaunchKernelGGL((k_soft_step_wide_fwd<22, true>), dim3(grid_for(n, 128)), dim3(128), 0, (hipStream_t)stream, logits,
                       x_soft, uniforms, x_next, flag, n, L, mask_index, temp, d_t, move_chance_s, 1.0f / gumbel_temp, seed,
                       step, sample_offset);
  else
    hipLaunchKernelGGL((k_soft_step_wide_fwd<22, false>), dim3(grid_for(n, 128)), dim3(128), 0, (hipStream_t)stream, logits,
                       x_soft, uniforms, x_next, flag, n, L, mask_index, temp, d_t, move_chance_s, 1.0f / gumbel_temp, seed,
                       step, sample_offset);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_reverse_step_soft_wide_bwd(const float* logits, const float* x_soft, const float* uniforms,
                                              const float* grad_x_next, const float* grad_flag, float* dlogits, float* dx,
                                              int B, int L, int V, int mask_index, float temp, float d_t, float move_chance_s,
                                              float gumbel_temp, uint64_t seed, uint32_t step, uint64_t sample_offset,
                                              void* stream) {
  DRK_SHAPE_GUARD("drk_reverse_step_soft_wide_bwd");
  ARGCHECK(logits && x_soft && grad_x_next && dlogits && dx && temp > 0.f && gumbel_temp > 0.f && V == 22 &&
               mask_index >= 0 && mask_index < V, "drk_reverse_step_soft_wide_bwd (V must be 22)");
  const long n = (long)B * L;
  ARGCHECK(n < (1L << 32), "drk_reverse_step_soft_wide_bwd (B*L must be below 2^32)");
  if (uniforms)
    hipLaunchKernelGGL((k_soft_step_wide_bwd<22, true>), dim3(grid_for(n, 128)), dim3(128), 0, (hipStream_t)stream, logits,
                       x_soft, uniforms, grad_x_next, grad_flag, dlogits, dx, n, L, mask_index, temp, d_t, move_chance_s,
                       1.0f / gumbel_temp, seed, step, sample_offset);
  else
    hipLaunchKernelGGL((k_soft_step_wide_bwd<22, false>), dim3(grid_for(n, 128)), dim3(128), 0, (hipStream_t)stream, logits,
                       x_soft, uniforms, grad_x_next, grad_flag, dlogits, dx, n, L, mask_index, temp, d_t, move_chance_s,
                       1.0f / gumbel_temp, seed, step, sample_offset);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
