// fp32 denoiser kernels (CNNModel.forward, models/dnaconv.py:74-110): the exact-arithmetic path
// used for parity with the fp32 reference and for the fine-tuning forward whose activations
// are kept for the backward pass.  Channels-last activations [B*L][128].
//
// The 128->128 k=9 dilated convolution is an implicit GEMM (M = B*L positions, N = 128,
// K = 9*128) on the f32-input MFMA v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain, at
// the fp32 vector peak (155 TF), with one VGPR per operand per lane.
#include <type_traits>
#include "drk_common.h"
#include "../../include/drakes_hip.h"

// ------------------------------------------------------------------------------------ time embedding
// One workgroup (128 threads) per batch row.  Argument of sin/cos reaches ~4e3 rad, so the
// accurate sinf/cosf are used (no fast-math intrinsics).
__global__ void __launch_bounds__(128) k_time_embed(const float* __restrict__ sigma, const float* __restrict__ W,
                                                    const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                                    float* __restrict__ temb) {
  __shared__ float four[DRK_C];
  const int b = blockIdx.x, c = threadIdx.x;
  const float s = sigma[b];
  const float proj = ((s * W[c & 63]) * 2.0f) * 3.14159274101257324f;  // x*W*2*np.pi, left to right in fp32
  four[c] = (c < 64) ? sinf(proj) : cosf(proj);
  __syncthreads();
  float acc = 0.0f;
  const float* wr = lin_w + (long)c * DRK_C;
#pragma unroll 8
  for (int j = 0; j < DRK_C; ++j) acc += wr[j] * four[j];
  acc += lin_b[c];
  temb[(long)b * DRK_C + c] = fmaxf(acc, 0.0f);
}

// grid (n_layers, nb), 128 threads: tbias[b][layer][c] = dense_w[layer][c][:] . temb[b][:] + dense_b[layer][c]
__global__ void __launch_bounds__(128) k_time_bias(const float* __restrict__ temb, const float* __restrict__ dense_w,
                                                   const float* __restrict__ dense_b, float* __restrict__ tbias,
                                                   int n_layers) {
  __shared__ float te[DRK_C];
  const int layer = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
  te[c] = temb[(long)b * DRK_C + c];
  __syncthreads();
  const float* wr = dense_w + ((long)layer * DRK_C + c) * DRK_C;
  float acc = 0.0f;
#pragma unroll 8
  for (int j = 0; j < DRK_C; ++j) acc += wr[j] * te[j];
  tbias[((long)b * n_layers + layer) * DRK_C + c] = acc + dense_b[layer * DRK_C + c];
}

// ------------------------------------------------------------------------------------ stem
// 16 flattened rows per workgroup; thread = (output channel, half of the rows).  The 45 weights
// of a channel sit in registers; the (one-hot or soft) inputs of rows r0-4 .. r0+19 in LDS.
#define STEM_ROWS 16
__global__ void __launch_bounds__(256) k_stem(const int64_t* __restrict__ x, const float* __restrict__ x_soft,
                                              const float* __restrict__ w_t, const float* __restrict__ bias,
                                              float* __restrict__ feat, long M, int L) {
  __shared__ float xs[(STEM_ROWS + 8) * DRK_V];
  const long r0 = (long)blockIdx.x * STEM_ROWS;
  for (int i = threadIdx.x; i < (STEM_ROWS + 8) * DRK_V; i += 256) {
    const long r = r0 - 4 + i / DRK_V;
    const int v = i % DRK_V;
    float val = 0.0f;
    if (r >= 0 && r < M) val = x ? ((int)x[r] == v ? 1.0f : 0.0f) : x_soft[r * DRK_V + v];
    xs[i] = val;
  }
  const int co = threadIdx.x & 127, sub = threadIdx.x >> 7;
  float w[DRK_TAPS * DRK_V];
#pragma unroll
  for (int k = 0; k < DRK_TAPS * DRK_V; ++k) w[k] = w_t[k * DRK_C + co];
  const float bco = bias[co];
  __syncthreads();
  for (int rr = sub * (STEM_ROWS / 2); rr < (sub + 1) * (STEM_ROWS / 2); ++rr) {
    const long r = r0 + rr;
    if (r >= M) break;
    const int p = (int)(r % L);
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < DRK_TAPS; ++t) {
      const int pp = p + t - 4;
      if (pp >= 0 && pp < L) {
#pragma unroll
        for (int v = 0; v < DRK_V; ++v) acc += xs[(rr + t) * DRK_V + v] * w[t * DRK_V + v];
      }
    }
    feat[r * DRK_C + co] = fmaxf(acc + bco, 0.0f);
  }
}

// Token input (the sampler's case): the one-hot picks ONE weight column per tap, so a row is nine 16-byte LDS reads
// and adds per lane instead of 45 multiply-adds — and the same bits: k_stem adds 1.0 * w at the token's column and
// 0.0 * w (= +-0) at the others, in the same tap order.  64 rows per workgroup, the 23 KB of weights LDS-resident;
// half-wave = one row (32 lanes x 4 channels), so the stores are whole 512-byte rows.
#define STEMT_ROWS 64
__global__ void __launch_bounds__(256) k_stem_tok(const int64_t* __restrict__ x, const float* __restrict__ w_t,
                                                  const float* __restrict__ bias, float* __restrict__ feat, long M, int L) {
  __shared__ __attribute__((aligned(16))) float wl[DRK_TAPS * DRK_V * DRK_C];   // [tap][v][co]
  __shared__ int toks[STEMT_ROWS + 8];
  const long r0 = (long)blockIdx.x * STEMT_ROWS;
  for (int i = threadIdx.x; i < DRK_TAPS * DRK_V * DRK_C / 4; i += 256) ((f32x4*)wl)[i] = ((const f32x4*)w_t)[i];
  if (threadIdx.x < STEMT_ROWS + 8) {
    const long r = r0 - 4 + threadIdx.x;
    toks[threadIdx.x] = (r >= 0 && r < M) ? (int)x[r] : -1;
  }
  const int cq = threadIdx.x & 31, rs = threadIdx.x >> 5;
  const f32x4 b4 = *(const f32x4*)(bias + 4 * cq);
  __syncthreads();
#pragma unroll 2
  for (int k = 0; k < STEMT_ROWS / 8; ++k) {
    const int rr = rs + 8 * k;
    const long r = r0 + rr;
    if (r >= M) break;
    const int p = (int)(r % L);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DRK_TAPS; ++t) {
      const int pp = p + t - 4;
      const int tk = toks[rr + t];
      if (pp >= 0 && pp < L && (unsigned)tk < (unsigned)DRK_V) acc += *(const f32x4*)(wl + (t * DRK_V + tk) * DRK_C + 4 * cq);
    }
    f32x4 o = acc + b4;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], 0.0f);
    *(f32x4*)(feat + r * DRK_C + 4 * cq) = o;
  }
}

// ------------------------------------------------------------------------------------ LayerNorm (+ time bias)
// 32 lanes per row (one float4 each), 8 rows per 256-thread workgroup.
__global__ void __launch_bounds__(256) k_ln_fwd(const float* __restrict__ feat, const float* __restrict__ tbias,
                                                long tb_stride, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float* __restrict__ h,
                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, long M,
                                                int L, int width) {
  // width < 128: a narrower model zero-padded into the 128 lanes (gamma = beta = 0 there): the statistics are taken over
  // the first `width` channels only; the other lanes hold 0 going in and coming out.
  const int lr = threadIdx.x & 31;
  const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= M) return;
  const long b = row / L;
  f32x4 v = *(const f32x4*)(feat + row * DRK_C + 4 * lr);
  if (tbias) v += *(const f32x4*)(tbias + b * tb_stride + 4 * lr);
  float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.0f / (float)width;
  const float mean = s * inv;
  const f32x4 d = v - mean;
  float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  if (4 * lr >= width) q = 0.0f;
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q * inv + 1e-5f);
  const f32x4 g = *(const f32x4*)(gamma + 4 * lr), be = *(const f32x4*)(beta + 4 * lr);
  *(f32x4*)(h + row * DRK_C + 4 * lr) = d * rstd * g + be;
  if (mean_out && lr == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// ------------------------------------------------------------------------------------ conv as implicit GEMM
// Workgroup = 4 waves = 32*MT rows x 128 output channels; wave w owns channels [32w, 32w+32).
// Per tap and per group of 8 input channels a lane loads one float4 of A per row tile (its row,
// channels 8g+4*half..+3) and one float4 of B (its output channel, same input channels), then
// issues 4 MFMAs per row tile.  Operands come straight from global memory: h rows are re-used by
// the 4 waves and the 9 taps through L1/L2, the 590 KB of weights stay L2-resident.
template <int MT, bool PF>
__global__ void __launch_bounds__(256) k_conv_f32(const float* __restrict__ h, const float* __restrict__ w_pk,
                                                  const float* __restrict__ bias, const float* resid, float* out,
                                                  unsigned char* __restrict__ relu_mask, long M, int L, int taps, int dil,
                                                  int relu) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int half = lane >> 5, lr = lane & 31;
  const long r0 = (long)blockIdx.x * (32 * MT);
  const int co = 32 * w + lr;
  f32x16 acc[MT];
  long rows[MT];
  int pos[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
    rows[mt] = r0 + 32 * mt + lr;
    pos[mt] = (int)(rows[mt] % L);
  }
  const int center = taps / 2;
  if (!PF) {
    for (int t = 0; t < taps; ++t) {
      const int off = (t - center) * dil;
      const float* src[MT];
      bool ok[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int p = pos[mt] + off;
        ok[mt] = rows[mt] < M && p >= 0 && p < L;
        src[mt] = h + (rows[mt] + off) * DRK_C + 4 * half;
      }
      const float* wsrc = w_pk + (((long)t * 32 + half) * DRK_C + co) * 4;
#pragma unroll 4
      for (int g = 0; g < 16; ++g) {
        const f32x4 b = *(const f32x4*)(wsrc + (long)g * (2 * DRK_C * 4));
        f32x4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          a[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (ok[mt]) a[mt] = *(const f32x4*)(src[mt] + 8 * g);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j], b[j], acc[mt], 0, 0, 0);
      }
    }
  } else {
    // K runs over (tap, 32-channel group): 4 groups per tap, each 4 x (8 channels = 4 MFMAs per row tile).  The operands
    // of group i+1 are loaded BEFORE the MFMAs of group i (ping-pong registers): one wave then keeps the MFMA pipe
    // busy through the L2 latency instead of idling between bursts, which is what small batches (one wave per SIMD) need.
    struct Group {
      f32x4 b[4];
      f32x4 a[MT][4];
    };
    auto load_group = [&](int idx, Group& G) {
      const int t = idx >> 2, gq = idx & 3;
      const int off = (t - center) * dil;
      const float* wsrc = w_pk + (((long)t * 32 + half) * DRK_C + co) * 4 + (long)(4 * gq) * (2 * DRK_C * 4);
  #pragma unroll
      for (int k = 0; k < 4; ++k) G.b[k] = *(const f32x4*)(wsrc + (long)k * (2 * DRK_C * 4));
  #pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int p = pos[mt] + off;
        const bool ok = rows[mt] < M && p >= 0 && p < L;
        const float* src = h + (rows[mt] + off) * DRK_C + 4 * half + 32 * gq;
  #pragma unroll
        for (int k = 0; k < 4; ++k) {
          G.a[mt][k] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (ok) G.a[mt][k] = *(const f32x4*)(src + 8 * k);
        }
      }
    };
    auto mma_group = [&](const Group& G) {
  #pragma unroll
      for (int k = 0; k < 4; ++k)
  #pragma unroll
        for (int j = 0; j < 4; ++j)
  #pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(G.a[mt][k][j], G.b[k][j], acc[mt], 0, 0, 0);
    };
    const int ngroups = 4 * taps;   // even
    Group g0, g1;
    load_group(0, g0);
    for (int idx = 0; idx < ngroups; idx += 2) {
      load_group(idx + 1, g1);
      mma_group(g0);
      if (idx + 2 < ngroups) load_group(idx + 2, g0);
      mma_group(g1);
    }
  }
  const float bco = bias ? bias[co] : 0.0f;
  // every residual value is requested before the first store: out may be resid, so a load left inside the store loop
  // could not be moved above the store before it and each would wait out its own round trip
  float rv[MT][16];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      long row = r0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * half;
      row = row < M ? row : M - 1;
      rv[mt][i] = resid ? resid[row * DRK_C + co] : 0.0f;
    }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long row = r0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * half;
      if (row < M) {
        float v = acc[mt][i] + bco;
        if (relu_mask) relu_mask[row * DRK_C + co] = v > 0.0f ? 1 : 0;   // kept for the backward pass
        if (relu) v = fmaxf(v, 0.0f);
        if (resid) v += rv[mt][i];
        out[row * DRK_C + co] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ conv, large batches
// Same implicit GEMM, 64 rows x 128 output channels per workgroup, but the activation tile of each tap goes through
// LDS.  In k_conv_f32 a lane's A fragment is 16 bytes of ITS row, so one wave-load touches 64 different 128-byte
// lines and the texture path, not the MFMA, paces the kernel (59 % of the f32 MFMA peak).  Here the 64 x 512-byte rows
// of a tap are fetched as whole rows (fully coalesced, 8 x 16 B per thread), parked in LDS with 528-byte rows
// (conflict-free ds_read_b128 for the fragment pattern), and the NEXT tap's rows are in flight in registers while the
// current tap's 128 MFMAs per wave run.  Weights still stream from L2 (coalesced, 590 KB per layer).
#define CL_ROWS 64
#define CL_STRIDE 132   // floats per LDS row (128 + 4 pad)
__global__ void __launch_bounds__(256) k_conv_f32_lds(const float* __restrict__ h, const float* __restrict__ w_pk,
                                                      const float* __restrict__ bias, const float* resid, float* out,
                                                      unsigned char* __restrict__ relu_mask, long M, int L, int taps, int dil,
                                                      int relu) {
  __shared__ __attribute__((aligned(16))) float tile[2][CL_ROWS * CL_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int half = lane >> 5, lr = lane & 31;
  const long r0 = (long)blockIdx.x * CL_ROWS;
  const int co = 32 * w + lr;
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;
  // staging role of this thread: rows srow + 8 i (i < 8), 16-byte chunk scol of each
  const int srow = tid >> 5, scol = tid & 31;
  int spos[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) spos[i] = (int)((r0 + srow + 8 * i) % L);
  const int center = taps / 2;
  f32x4 stage[8];
  unsigned live = 0;
  // Unconditional loads from in-range rows, blanked when parked: the number of loads in flight is then static, and the
  // compiler's waits for the weight fragments do not also wait for the rows requested after them.
  auto fetch = [&](int t) {
    const int off = (t - center) * dil;
    live = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long row = r0 + srow + 8 * i;
      const int p = spos[i] + off;
      const bool ok = row < M && p >= 0 && p < L;
      live |= ok ? (1u << i) : 0u;
      const long src = ok ? row + off : r0;            // a dead row re-reads the tile's first row: a cache hit
      stage[i] = *(const f32x4*)(h + src * DRK_C + 4 * scol);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *(f32x4*)(&tile[buf][(srow + 8 * i) * CL_STRIDE + 4 * scol]) = (live >> i) & 1 ? stage[i] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  fetch(0);
  park(0);
  __syncthreads();
  // One tap: the first four weight fragments are requested BEFORE the next tap's rows — loads return in order, so a
  // fragment requested behind the rows would make the tap's first MFMA wait out their round trip.
  auto one_tap = [&](const int t, auto more_c) {
    constexpr bool more = decltype(more_c)::value;
    const int buf = t & 1;
    const float* wsrc = w_pk + (((long)t * 32 + half) * DRK_C + co) * 4;
    f32x4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *(const f32x4*)(wsrc + (long)g * (2 * DRK_C * 4));
    if constexpr (more) fetch(t + 1);             // in flight under this tap's MFMAs
    const float* a0 = &tile[buf][lr * CL_STRIDE + 4 * half];
    const float* a1 = a0 + 32 * CL_STRIDE;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 b = g < 4 ? bq[g] : *(const f32x4*)(wsrc + (long)g * (2 * DRK_C * 4));
      const f32x4 x0 = *(const f32x4*)(a0 + 8 * g);
      const f32x4 x1 = *(const f32x4*)(a1 + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[j], b[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[j], b[j], acc[1], 0, 0, 0);
      }
    }
    if constexpr (more) {
      park(buf ^ 1);                              // the other buffer: its last readers finished before the previous barrier
      __syncthreads();
    }
  };
  for (int t = 0; t + 1 < taps; ++t) one_tap(t, std::true_type{});
  one_tap(taps - 1, std::false_type{});
  const float bco = bias ? bias[co] : 0.0f;
  float rv[2][16];          // the residual, requested before the first store (see k_conv_f32)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      long row = r0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * half;
      row = row < M ? row : M - 1;
      rv[mt][i] = resid ? resid[row * DRK_C + co] : 0.0f;
    }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long row = r0 + 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * half;
      if (row < M) {
        float v = acc[mt][i] + bco;
        if (relu_mask) relu_mask[row * DRK_C + co] = v > 0.0f ? 1 : 0;
        if (relu) v = fmaxf(v, 0.0f);
        if (resid) v += rv[mt][i];
        out[row * DRK_C + co] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ final 128->5 projection
// 8 lanes per row; lane s takes float4 chunks s, s+8, s+16, s+24 of the row (coalesced), reduces
// over the 8 lanes by shuffles; lanes 0..4 of the group store one logit each.
__global__ void __launch_bounds__(256) k_head_out(const float* __restrict__ y, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ logits, long M) {
  __shared__ float ws[DRK_V * DRK_C];
  for (int i = threadIdx.x; i < DRK_V * DRK_C; i += 256) ws[i] = w[i];
  __syncthreads();
  const int s = threadIdx.x & 7;
  const long row = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
  float part[DRK_V] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (row < M) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ch = 4 * (s + 8 * c);
      const f32x4 a = *(const f32x4*)(y + row * DRK_C + ch);
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) {
        const f32x4 ww = *(const f32x4*)(ws + v * DRK_C + ch);
        part[v] += (a[0] * ww[0] + a[1] * ww[1]) + (a[2] * ww[2] + a[3] * ww[3]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) {
    part[v] += __shfl_xor(part[v], 4);
    part[v] += __shfl_xor(part[v], 2);
    part[v] += __shfl_xor(part[v], 1);
  }
  if (row < M && s < DRK_V) {
    float o = part[0];
#pragma unroll
    for (int v = 1; v < DRK_V; ++v) o = (s == v) ? part[v] : o;
    logits[row * DRK_V + s] = o + bias[s];
  }
}

// ------------------------------------------------------------------------------------ C ABI
#define ARGCHECK(cond, name) \
  if (!(cond)) return drk_set_error_msg(DRK_EINVAL, name ": bad arguments")

extern "C" int drk_time_embed_fwd(const float* sigma, int nb, const float* W, const float* lin_w, const float* lin_b,
                                  float* temb, void* stream) {
  ARGCHECK(sigma && W && lin_w && lin_b && temb && nb >= 0, "drk_time_embed_fwd");
  if (nb == 0) return DRK_OK;
  hipLaunchKernelGGL(k_time_embed, dim3(nb), dim3(128), 0, (hipStream_t)stream, sigma, W, lin_w, lin_b, temb);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_time_bias_fwd(const float* temb, int nb, const float* dense_w, const float* dense_b, int n_layers,
                                 float* tbias, void* stream) {
  ARGCHECK(temb && dense_w && dense_b && tbias && nb >= 0 && n_layers > 0, "drk_time_bias_fwd");
  if (nb == 0) return DRK_OK;
  hipLaunchKernelGGL(k_time_bias, dim3(n_layers, nb), dim3(128), 0, (hipStream_t)stream, temb, dense_w, dense_b, tbias,
                     n_layers);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_stem_fwd(const int64_t* x, const float* x_soft, const float* w_t, const float* bias, float* feat,
                            int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_stem_fwd");
  ARGCHECK((x || x_soft) && w_t && bias && feat && B >= 0 && L > 0, "drk_stem_fwd");
  const long M = (long)B * L;
  if (M == 0) return DRK_OK;
  if (x)
    hipLaunchKernelGGL(k_stem_tok, dim3((unsigned)((M + STEMT_ROWS - 1) / STEMT_ROWS)), dim3(256), 0, (hipStream_t)stream, x,
                       w_t, bias, feat, M, L);
  else
    hipLaunchKernelGGL(k_stem, dim3((unsigned)((M + STEM_ROWS - 1) / STEM_ROWS)), dim3(256), 0, (hipStream_t)stream, x,
                       x_soft, w_t, bias, feat, M, L);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_ln_fwd(const float* feat, const float* tbias, long tbias_bstride, const float* gamma,
                          const float* beta, float* h, float* mean, float* rstd, int B, int L, int width, void* stream) {
  DRK_SHAPE_GUARD("drk_ln_fwd");
  ARGCHECK(feat && gamma && beta && h && B >= 0 && L > 0 && (!mean == !rstd), "drk_ln_fwd");
  ARGCHECK(width >= 4 && width <= DRK_C && width % 4 == 0, "drk_ln_fwd (width: a multiple of 4 in [4, 128])");
  const long M = (long)B * L;
  if (M == 0) return DRK_OK;
  hipLaunchKernelGGL(k_ln_fwd, dim3((unsigned)((M + 7) / 8)), dim3(256), 0, (hipStream_t)stream, feat, tbias,
                     tbias_bstride, gamma, beta, h, mean, rstd, M, L, width);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_conv_fwd_f32(const float* h, const float* w_pk, const float* bias, const float* resid, float* out,
                                unsigned char* relu_mask, int B, int L, int taps, int dilation, int relu, void* stream) {
  DRK_SHAPE_GUARD("drk_conv_fwd_f32");
  ARGCHECK(h && w_pk && out && B >= 0 && L > 0 && taps >= 1 && (taps & 1) && dilation >= 0, "drk_conv_fwd_f32");
  const long M = (long)B * L;
  if (M == 0) return DRK_OK;
  if (M >= 64L * 256)        // one 64-row workgroup per CU or more: activation rows staged through LDS (measured faster
                             // from 25 600 rows up; at 6 400 rows the prefetching 32-row kernel below is 1.5x faster)
    hipLaunchKernelGGL(k_conv_f32_lds, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, h, w_pk,
                       bias, resid, out, relu_mask, M, L, taps, dilation, relu);
  else if (M > 32L * 384)    // 32-row tiles, twice the workgroups
    hipLaunchKernelGGL((k_conv_f32<1, false>), dim3((unsigned)((M + 31) / 32)), dim3(256), 0, (hipStream_t)stream, h, w_pk,
                       bias, resid, out, relu_mask, M, L, taps, dilation, relu);
  else                       // at most ~1.5 workgroups per CU: software-prefetched operands
    hipLaunchKernelGGL((k_conv_f32<1, true>), dim3((unsigned)((M + 31) / 32)), dim3(256), 0, (hipStream_t)stream, h, w_pk,
                       bias, resid, out, relu_mask, M, L, taps, dilation, relu);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_head_out_fwd(const float* y, const float* w, const float* bias, float* logits, int B, int L,
                                void* stream) {
  DRK_SHAPE_GUARD("drk_head_out_fwd");
  ARGCHECK(y && w && bias && logits && B >= 0 && L > 0, "drk_head_out_fwd");
  const long M = (long)B * L;
  if (M == 0) return DRK_OK;
  hipLaunchKernelGGL(k_head_out, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, (hipStream_t)stream, y, w, bias, logits,
                     M);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
