// Fused one-sequence-per-workgroup kernels of the differentiable bf16 path (round 3): see the two block comments below.
//
// Saved-tensor formats.  What only these two kernels exchange is kept in THEIR lane order, so that every wave-load /
// wave-store is one contiguous, fully coalesced piece (row-major pieces of 16 rows x 64 bytes cost 2-3x as much here):
//   featn  float32 [n_layers][B][8 waves][14 slots][64 lanes][4]: residual snapshot AFTER layer i in block i; wave
//          w = ct + 4 grp, slot 2 t + s, lane 16 g + n holds position 16 (grp + 2 t) + pn(n), channels 32 ct + 16 s + 4 g + 0..3
//          (the accumulator layout of drk_fused_common.h); DRK_FEATN_SEQ = 28 672 floats per sequence and layer;
//   maskp  uint64 [n_layers][B][8 waves][64 lanes]: the ReLU gates of layer i, bit 8 t + 4 s + j for channel + j.
// Row-major as everywhere else: feat0 (the stem's output, float32 [B*L][128]), hs (bf16 [n_layers+1][B*L][128]: the
// weight gradients' operand), dcs, means / rstds, y, dpre.  drakes_amd/fastpath.py::native_feat_index / unpack_masks map
// the two formats onto rows for the tests.
#include "drk_fused_common.h"

#define DRK_FEATN_SEQ (8 * 14 * 256)     // floats per sequence and layer in featn
#define DRK_MASKP_SEQ (8 * 64)           // uint64 per sequence and layer in maskp

// =====================================================================================================================
// k_train_fwd_bf16 — ONE differentiable denoiser forward per launch on the same body (round 3): the residual layers and
// final_conv.0 of CNNModel.forward (models/dnaconv.py:88-101) for the Gumbel-softmax steps of
// _sample_finetune_gradient (diffusion_gosai_update.py:552-630), one sequence per workgroup, the residual stream in
// registers, and everything the backward pass needs streamed out on the way: the residual snapshots (fp32), the
// LayerNorm outputs (bf16, the weight gradients' operand), the ReLU gates and the LayerNorm statistics.  It replaces
// drk_ln_fwd_bf16 + n_layers x drk_layer_bf16 + drk_layer_bf16(final_conv.0): 22 launches of ~20 us at fine-tune batch
// sizes.  The stem (fp32, exact) runs before it, the 128 -> 5 conv (fp32, exact) after it, as on the layer path.
struct TrainArgs {
  const char* wfrag;        // the fragment stream (fastpath.py::pack_stream)
  const float* vecs;        // [n_layers+1][3][128] (see SampleArgs)
  const float* tbias;       // [n_layers][128], shared by the batch
  const float* feat0;       // [M][128] fp32: the stem's output
  float* featn;             // lane order, see above
  char* hs;                 // [n_layers+1][M][128] bf16: LayerNorm outputs; the last block = bf16 of the last residual state
  unsigned long long* maskp;   // lane order, see above
  float* means;             // [n_layers][M]
  float* rstds;
  float* y;                 // [M][128] fp32: relu(final_conv.0)
  int B, L, n_layers;
  unsigned long long dilpk[4];
};

// The image's rows 0..L-1 (the LayerNorm output the layer just consumed, or dc) -> a row-major bf16 block: whole
// 256-byte rows in 16-byte pieces, every wave-store 1 KB contiguous.
__device__ __forceinline__ void image_rows_out(const char* smem, char* dst, int L, int tid) {
  for (int i = tid; i < L * 16; i += 512)
    *(f32x4*)(dst + (size_t)i * 16) = *(const f32x4*)(smem + OFF_HB + (i >> 4) * ROWB + 16 * (i & 15));
}

// epilogue_ln with the saves.  ACCZERO: LayerNorm of feat as it is (the first layer's input), nothing of acc is used.
// feat_o / mask_o: this WAVE's piece of featn / maskp for the layer just computed; hprev_o: where the image's current
// content (the LayerNorm output this layer consumed) goes, row-major; do_ln with mean_o / rstd_o: the next LayerNorm.
template <int NT, bool ACCZERO, bool FAST>
__device__ __forceinline__ void epilogue_ln_train(char* smem, f32x4 (&feat)[NT][2], const f32x4 (&acc)[NT][2], const float* blk,
                                                  const float* tbv_src, int L, int lane, int ct, int grp, int tid, float* feat_o,
                                                  unsigned long long* mask_o, char* hprev_o, bool do_ln, float* mean_o,
                                                  float* rstd_o) {
#pragma clang fp contract(fast)
  const int lo = opaque(lane);
  const int n = lo & 15, g = lo >> 4;
  const int pn = n ^ ((n & 8) >> 1);
  const int ch0 = 32 * ct + 4 * g;
  f32x4 tbv[2];
  if (do_ln) load_vec8(tbv_src, ch0, tbv);
  auto relu4 = [](f32x4 v) __attribute__((always_inline)) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 z = {0, 0, 0, 0};
    return __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, v), z));
  };
  if constexpr (!ACCZERO) {
    unsigned long long pm = 0ull;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 c = acc[t][s];
        feat[t][s] += relu4(c);
        const unsigned nib = (c[0] > 0.f ? 1u : 0u) | (c[1] > 0.f ? 2u : 0u) | (c[2] > 0.f ? 4u : 0u) | (c[3] > 0.f ? 8u : 0u);
        pm |= (unsigned long long)nib << (8 * t + 4 * s);
        *(f32x4*)(feat_o + (2 * t + s) * 256 + (size_t)(lo * 4)) = feat[t][s];     // (positions >= L: never read back)
      }
    }
    mask_o[lo] = pm;
  }
  if (hprev_o) image_rows_out(smem, hprev_o, L, tid);
  if (!do_ln) return;
  float* part = (float*)(smem + OFF_PART);
  float y[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 x0 = feat[t][0] + tbv[0], x1 = feat[t][1] + tbv[1];
    const f32x4 sv = x0 + x1;
    const f32x4 qv = x0 * x0 + x1 * x1;
    const float s1 = (sv[0] + sv[1]) + (sv[2] + sv[3]);
    const float s2 = (qv[0] + qv[1]) + (qv[2] + qv[3]);
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s2), false, false);
    y[t] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const int stat_ct = (g >> 1) * 4 + ct;
#pragma unroll
  for (int t = 0; t < NT; t += 2) {
    const int tb = (t + 1 < NT) ? t + 1 : t;
    const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[t]), __float_as_uint(y[tb]), false, false);
    const int tile = (g & 1) ? tb : t;
    part[(16 * (grp + 2 * tile) + pn) * PART_STRIDE + stat_ct] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  f32x4 gv[2], bv[2];
  load_vec8(blk + 128, ch0, gv);
  load_vec8(blk + 256, ch0, bv);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = 16 * (grp + 2 * t) + pn;
    const f32x4 q1 = *(const f32x4*)(part + p * PART_STRIDE), q2 = *(const f32x4*)(part + p * PART_STRIDE + 4);
    const float s1 = (q1[0] + q1[1]) + (q1[2] + q1[3]);
    const float s2 = (q2[0] + q2[1]) + (q2[2] + q2[3]);
    const float mean = s1 * (1.0f / 128.0f);
    const float var = fmaxf(s2 * (1.0f / 128.0f) - mean * mean, 0.0f);
    const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
    const float c = -mean * rstd;
    f32x4 hv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) hv[s][i] = ((feat[t][s][i] + tbv[s][i]) * rstd + c) * gv[s][i] + bv[s][i];
    if ((FAST && grp + 2 * t < 12) || p < L) {
      store_h8(smem, p, ch0, hv);
      if (ct == 0 && g == 0) {
        mean_o[p] = mean;
        rstd_o[p] = rstd;
      }
    }
  }
  __syncthreads();
}

template <int NT, int GRP, int NTHR, bool FAST>
__device__ __forceinline__ void train_fwd_body(const TrainArgs& a, char* smem, int tid, int wave, int lane) {
  const int ct = wave & 3;
  constexpr int grp = GRP;
  const int L = a.L;
  const long b = blockIdx.x;
  const long M = (long)a.B * L;
  const float* vecs = (const float*)(smem + OFF_VECS);
  const float* tbl = (const float*)(smem + OFF_TB);
  auto dil_of = [&](int layer) { return (int)((a.dilpk[layer >> 3] >> (8 * (layer & 7))) & 255ull); };
  auto reach_of = [&](int layer) {
    const int d = dil_of(layer);
    if (FAST) return d == 64 ? (GRP == 1 ? 2 : 3) : 4;
    const int m = (d > 0) ? (L - 1) / d : 4;
    return m < 4 ? m : 4;
  };
  const char* wuni = a.wfrag + ct * (2 * SUB_BYTES);
  const unsigned loff = (unsigned)lane * 16u;
  const char* wlane = wuni + loff;
  const int head0_tap = 9 * a.n_layers;
  const int first_tap = 4 - reach_of(0);

  for (int i = tid; i < (HPAD + ZROW + 1) * ROWB / 16; i += NTHR) ((f32x4*)smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < (a.n_layers + 1) * 3 * 128 / 4; i += NTHR) ((f32x4*)(smem + OFF_VECS))[i] = ((const f32x4*)a.vecs)[i];
  for (int i = tid; i < a.n_layers * 128 / 4; i += NTHR) ((f32x4*)(smem + OFF_TB))[i] = ((const f32x4*)a.tbias)[i];
  bf16v8 ring[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
    ring[f] = *(const bf16v8*)(wlane + (long)first_tap * TAP_BYTES + (f >> 1) * KSTEP_BYTES + (f & 1) * SUB_BYTES);

  f32x4 feat[NT][2], acc[NT][2];
  {   // the stem's output: this lane's 8 channels of its positions
    const int lo = opaque(lane);
    const int pn = (lo & 15) ^ ((lo & 8) >> 1);
    const int ch0 = 32 * ct + 4 * (lo >> 4);
    const float* f0 = a.feat0 + b * L * 128;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int p = 16 * (grp + 2 * t) + pn;
      const bool ok = p < L;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      feat[t][0] = ok ? *(const f32x4*)(f0 + p * 128 + ch0) : z;
      feat[t][1] = ok ? *(const f32x4*)(f0 + p * 128 + ch0 + 16) : z;
    }
  }
  __syncthreads();   // image zeroed, tables resident
  epilogue_ln_train<NT, true, FAST>(smem, feat, acc, vecs, tbl, L, lane, ct, grp, tid, nullptr, nullptr, nullptr, true,
                                    a.means + b * L, a.rstds + b * L);
  const int wv = ct + 4 * grp;

  int m_next = reach_of(0);
  for (int layer = 0; layer < a.n_layers; ++layer) {
    const int dil = dil_of(layer);
    const int m = m_next;
    m_next = (layer + 1 < a.n_layers) ? reach_of(layer + 1) : 0;
    const int after = (layer + 1 < a.n_layers) ? 9 * (layer + 1) + 4 - m_next : head0_tap;
    {
      f32x4 cbl[2];
      load_vec8(vecs + 384 * (layer + 1), 32 * ct + 4 * (opaque(lane) >> 4), cbl);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t][0] = cbl[0];
        acc[t][1] = cbl[1];
      }
    }
    const char* wn = (FAST ? wuni : wlane) + (long)(9 * layer + 4 - m + 1) * TAP_BYTES;
    const char* wafter = (FAST ? wuni : wlane) + (long)after * TAP_BYTES;
    if constexpr (FAST) {
      if (dil == 64) {
        static_layer<NT, GRP, 64, (GRP == 1 ? 2 : 3)>(smem, wn, wafter, loff, ring, lane, acc);
      } else if (dil == 16) {
        static_layer<NT, GRP, 16, 4>(smem, wn, wafter, loff, ring, lane, acc);
      } else {
        const char* hcur = pad_base(smem, lane) - 4 * dil * ROWB;
        const int hstep = dil * ROWB;
        bf16v8 cur[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(hcur + RI(GRP, t, 0));
        for (int k = 0; k < 8; k += 2) {
          TAP_PRIO(GRP == 0);
          mma_tap_pad<NT, GRP, true>(hcur, hcur + hstep, wn, loff, ring, cur, acc);
          hcur += hstep;
          wn += TAP_BYTES;
          TAP_PRIO(GRP != 0);
          mma_tap_pad<NT, GRP, true>(hcur, hcur + hstep, wn, loff, ring, cur, acc);
          hcur += hstep;
          wn += TAP_BYTES;
        }
        TAP_PRIO(GRP == 0);
        mma_tap_pad<NT, GRP, false>(hcur, hcur, wafter, loff, ring, cur, acc);
      }
    } else {
      const char* bp[NT];
      bf16v8 cur[NT];
      int off = -m * dil;
      tap_begin<NT>(smem, off, L, lane, grp, bp, cur);
      for (int k = 0; k < 2 * m; ++k) {
        off += dil;
        mma_tap<NT, true>(smem, wn, ring, off, L, lane, grp, bp, cur, acc);
        wn += TAP_BYTES;
      }
      mma_tap<NT, false>(smem, wafter, ring, 0, L, lane, grp, bp, cur, acc);
    }
    TAP_PRIO(false);
    const bool ln_next = layer + 1 < a.n_layers;
    const long rows1 = ((long)(layer + 1) * M + b * L);
    const long seq = (long)layer * a.B + b;
    epilogue_ln_train<NT, false, FAST>(smem, feat, acc, vecs + 384 * (layer + 1), tbl + 128 * (layer + 1), L, lane, ct, grp, tid,
                                       a.featn + seq * DRK_FEATN_SEQ + wv * (14 * 256), a.maskp + seq * DRK_MASKP_SEQ + wv * 64,
                                       a.hs + ((long)layer * M + b * L) * 256, ln_next, a.means + rows1, a.rstds + rows1);
  }

  // ---- final_conv.0 (1x1) + ReLU on bf16(last residual state), which is also the weight gradient's operand hs[n] ----
  __syncthreads();   // the last layer did no LayerNorm: every wave is past its reads of the h image
  {
    const int lo_h = opaque(lane);
    const int pn_h = (lo_h & 15) ^ ((lo_h & 8) >> 1);
    const int ch_h = 32 * ct + 4 * (lo_h >> 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int p = 16 * (grp + 2 * t) + pn_h;
      if ((FAST && grp + 2 * t < 12) || p < L) store_h8(smem, p, ch_h, feat[t]);
      acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  if constexpr (FAST) {
    const char* hcur = pad_base(smem, lane);
    bf16v8 cur[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(hcur + RI(GRP, t, 0));
    mma_tap_pad<NT, GRP, false>(hcur, hcur, wuni + (long)head0_tap * TAP_BYTES, loff, ring, cur, acc);   // (the refill is unused)
  } else {
    const char* bp[NT];
    bf16v8 cur[NT];
    tap_begin<NT>(smem, 0, L, lane, grp, bp, cur);
    mma_tap<NT, false>(smem, wlane + (long)head0_tap * TAP_BYTES, ring, 0, L, lane, grp, bp, cur, acc);
  }
  image_rows_out(smem, a.hs + ((long)a.n_layers * M + b * L) * 256, L, tid);   // hs[n]: the image is stable since the barrier
  {
    const int lo2 = opaque(lane);
    const int pn2 = (lo2 & 15) ^ ((lo2 & 8) >> 1);
    const int ch2 = 32 * ct + 4 * (lo2 >> 4);
    f32x4 hb0[2];
    load_vec8(vecs + 384 * a.n_layers + 128, ch2, hb0);
    float* yo = a.y + b * L * 128;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int p = 16 * (grp + 2 * t) + pn2;
      if (p < L) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          f32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = fmaxf(acc[t][s][i] + hb0[s][i], 0.f);
          *(f32x4*)(yo + (long)p * 128 + ch2 + 16 * s) = o;
        }
      }
    }
  }
}

template <bool FAST>
__global__ void __launch_bounds__(512, 2) k_train_fwd_bf16(const TrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4)
    train_fwd_body<7, 0, 512, FAST>(a, smem, tid, wave, lane);
  else if (FAST || a.L <= 208)
    train_fwd_body<6, 1, 512, FAST>(a, smem, tid, wave, lane);
  else
    train_fwd_body<7, 1, 512, false>(a, smem, tid, wave, lane);
}

// =====================================================================================================================
// k_train_bwd_bf16 — the backward of those layers in ONE launch, on the same body run in reverse: per layer the
// convolution's input gradient (the forward's MMA phase on the transposed, tap-flipped stream, fastpath.py::
// pack_stream_T), the LayerNorm backward, the residual add and the ReLU gate of the layer below — what
// drk_layer_bwd_bf16 does per launch — with the residual-GRADIENT stream in registers and the gated gradient dc (the next
// convolution's operand, and the weight gradients' operand) in the LDS image.  Reads the forward's saved blocks.
#define OFF_STAT OFF_LOG          // float [2][224]: mean / rstd of the layer being processed (the sampler's logit area)
struct TrainBwdArgs {
  const char* wfrag_t;      // the transposed stream
  const float* vecs;        // [n_layers+1][3][128] (gamma_i = row 3 i + 1)
  const float* tbias;       // [n_layers][128]
  const float* feat0;       // [M][128]: the stem's output
  const float* featn;       // the forward's residual snapshots, lane order (see the head of this file)
  const unsigned long long* maskp;   // the forward's ReLU gates, lane order
  const float* means;       // [n_layers][M]
  const float* rstds;
  char* dcs;                // bf16 [slots][M][128]; slot (n_layers mod slots) holds final_conv.0's gated output gradient (input)
  int slots;                // n_layers + 1: every dc kept; 2: ping-pong
  float* dpre;              // [M][128] fp32 out: dL/d(stem pre-activation) = dL/dfeat_0 * (feat_0 > 0)
  float* partials;          // [2 B] slabs of part_stride floats (slab 2 b + position group), slot i at 512 i; or NULL
  long part_stride;
  int B, L, n_layers;
  unsigned long long dilpk[4];
};

// The epilogue of one backward layer.  acc = dh (the convolution's input gradient).  LN: the LayerNorm backward
//     xhat = (feat + tbias - mean) * rstd;  dyg = dh * gamma;  dx = rstd * (dyg - mean_c(dyg) - xhat * mean_c(dyg * xhat))
// else dx = dh (final_conv.0 has no LayerNorm in front of it).  gr += dx.  If mask_n is given: dc_next = bf16(gr * gate)
// -> the LDS image (the next convolution's operand).  Column sums -> this wave's partial slab.
// Register budget (gr, acc and the prefetched weight ring are live throughout: 144): the LayerNorm's input is requested
// here, not under the taps, and folded away tile by tile — after pass 1 gr holds gr + rstd * dyg and acc holds
// rstd * xhat, so pass 2 is gr -= rstd * m1 + acc * m2 and nothing else of the tile is kept; the per-channel vectors are
// re-read from LDS per tile; dgamma / dbeta leave before the barrier.
template <int NT, bool LN, bool FAST>
__device__ __forceinline__ void epilogue_lnb(char* smem, f32x4 (&gr)[NT][2], f32x4 (&acc)[NT][2], const float* feat_i, bool feat_rows,
                                             const float* gamma, const float* tbv_src, int L, int lane, int ct, int grp,
                                             const unsigned long long* mask_n, float* part_o, const float* mean_nx,
                                             const float* rstd_nx, int tid, char* dc_o, const float* stat, float* stat_nx,
                                             bf16v8 (&ring)[8], const char* wnext) {
#pragma clang fp contract(fast)
  const int lo = opaque(lane);
  const int n = lo & 15, g = lo >> 4;
  const int pn = n ^ ((n & 8) >> 1);
  const int ch0 = 32 * ct + 4 * g;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* part = (float*)(smem + OFF_PART);
  auto row_sum16 = [](float v) __attribute__((always_inline)) {   // over the 16 lanes of a DPP row, fixed order
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, false));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, false));
    return v;
  };
  auto put_sum = [&](f32x4 v0, f32x4 v1, int q) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v0[i] = row_sum16(v0[i]);
      v1[i] = row_sum16(v1[i]);
    }
    if (n == 0) {
      *(f32x4*)(part_o + ch0 + q * 128) = v0;
      *(f32x4*)(part_o + ch0 + q * 128 + 16) = v1;
    }
  };
  f32x4 sx0 = zero4, sx1 = zero4;
  if constexpr (LN) {
    f32x4 xh[NT][2];
    // (requested here, not under the taps: 56 more live registers there spill, and in flight under the taps they were
    // measured no faster)  feat_rows: layer 0's input is the stem's row-major output; otherwise this wave's piece of featn
    if (feat_rows) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p = 16 * (grp + 2 * t) + pn;
        const bool ok = (FAST && grp + 2 * t < 12) || p < L;
        xh[t][0] = ok ? *(const f32x4*)(feat_i + (size_t)(p * 128 + ch0)) : zero4;
        xh[t][1] = ok ? *(const f32x4*)(feat_i + (size_t)(p * 128 + ch0 + 16)) : zero4;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        xh[t][0] = *(const f32x4*)(feat_i + (2 * t) * 256 + (size_t)(lo * 4));
        xh[t][1] = *(const f32x4*)(feat_i + (2 * t + 1) * 256 + (size_t)(lo * 4));
      }
    }
    f32x4 sg0 = zero4, sg1 = zero4, sb0 = zero4, sb1 = zero4;
    float y[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int p = 16 * (grp + 2 * t) + pn;
      const bool ok = (FAST && grp + 2 * t < 12) || p < L;
      const float mu = stat[p], rs = stat[MAXPOS + p];
      f32x4 tbv[2], gm[2];
      load_vec8(tbv_src, opaque(ch0), tbv);
      load_vec8(gamma, opaque(ch0), gm);
      const f32x4 d0 = ok ? acc[t][0] : zero4, d1 = ok ? acc[t][1] : zero4;
      const f32x4 h0 = (xh[t][0] + tbv[0] - mu) * rs, h1 = (xh[t][1] + tbv[1] - mu) * rs;
      sg0 += d0 * h0;
      sg1 += d1 * h1;
      sb0 += d0;
      sb1 += d1;
      const f32x4 y0 = d0 * gm[0], y1 = d1 * gm[1];
      const f32x4 sv = y0 + y1, qv = y0 * h0 + y1 * h1;
      const float s1 = (sv[0] + sv[1]) + (sv[2] + sv[3]);
      const float s2 = (qv[0] + qv[1]) + (qv[2] + qv[3]);
      const f32x4 a0 = y0 * rs, a1 = y1 * rs;
      gr[t][0] += a0;
      gr[t][1] += a1;
      sx0 += a0;
      sx1 += a1;
      acc[t][0] = h0 * rs;
      acc[t][1] = h1 * rs;
      const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s2), false, false);
      y[t] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const int stat_ct = (g >> 1) * 4 + ct;
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
      const int tb = (t + 1 < NT) ? t + 1 : t;
      const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[t]), __float_as_uint(y[tb]), false, false);
      const int tile = (g & 1) ? tb : t;
      part[(16 * (grp + 2 * tile) + pn) * PART_STRIDE + stat_ct] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (part_o) {
      put_sum(sg0, sg1, 0);
      put_sum(sb0, sb1, 1);
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int p = 16 * (grp + 2 * t) + pn;
      const bool ok = (FAST && grp + 2 * t < 12) || p < L;
      const f32x4 d0 = ok ? acc[t][0] : zero4, d1 = ok ? acc[t][1] : zero4;
      gr[t][0] += d0;
      gr[t][1] += d1;
      sx0 += d0;
      sx1 += d1;
    }
    if (part_o) {
      put_sum(zero4, zero4, 0);
      put_sum(zero4, zero4, 1);
    }
  }
  // The next layer's first weight fragments are requested HERE, not during this layer's last tap (whose refill the
  // caller points at them too: the compiler drops those loads, their results are overwritten unread): the ring's 32
  // registers are free during pass 1.  Ahead of the dc stores below, so that the next tap's wait does not cover them.
#pragma unroll
  for (int f = 0; f < 8; ++f) ring[f] = *(const bf16v8*)(wnext + (f >> 1) * KSTEP_BYTES + (f & 1) * SUB_BYTES);
  // the gate of the layer below, and the statistics of ITS LayerNorm (used by the next call): requested ahead of the barrier
  unsigned long long pm = 0ull;
  if (mask_n) pm = mask_n[lo];
  float st_m = 0.f, st_r = 0.f;
  if (mean_nx && tid < L) {
    st_m = mean_nx[tid];
    st_r = rstd_nx[tid];
  }
  if (dc_o) image_rows_out(smem, dc_o, L, tid);   // this layer's dc (the image until the barrier) -> its block, for the weight gradient
  __syncthreads();   // partials visible; every wave is past its reads of the dc image
  f32x4 sc0 = zero4, sc1 = zero4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = 16 * (grp + 2 * t) + pn;
    if constexpr (LN) {
      const float* pp = part + p * PART_STRIDE;
      const f32x4 q1 = *(const f32x4*)(pp), q2 = *(const f32x4*)(pp + 4);
      const float m1 = ((q1[0] + q1[1]) + (q1[2] + q1[3])) * (1.0f / 128.0f);
      const float m2 = ((q2[0] + q2[1]) + (q2[2] + q2[3])) * (1.0f / 128.0f);
      const float c1 = stat[MAXPOS + p] * m1;
      const f32x4 e0 = acc[t][0] * m2 + c1, e1 = acc[t][1] * m2 + c1;
      const bool ok = (FAST && grp + 2 * t < 12) || p < L;      // (rows >= L: acc holds rstd * xhat of a zero row, dyg = 0, m1 = m2 = 0)
      gr[t][0] -= ok ? e0 : zero4;
      gr[t][1] -= ok ? e1 : zero4;
      sx0 -= ok ? e0 : zero4;
      sx1 -= ok ? e1 : zero4;
    }
    if (mask_n) {
      const bool okm = (FAST && grp + 2 * t < 12) || p < L;
      const unsigned m8 = okm ? (unsigned)(pm >> (8 * t)) & 255u : 0u;
      const f32x4 o0 = gr[t][0], o1 = gr[t][1];
      f32x4 v[2];
      v[0] = f32x4{(m8 & 1u) ? o0[0] : 0.f, (m8 & 2u) ? o0[1] : 0.f, (m8 & 4u) ? o0[2] : 0.f, (m8 & 8u) ? o0[3] : 0.f};
      v[1] = f32x4{(m8 & 16u) ? o1[0] : 0.f, (m8 & 32u) ? o1[1] : 0.f, (m8 & 64u) ? o1[2] : 0.f, (m8 & 128u) ? o1[3] : 0.f};
      sc0 += v[0];
      sc1 += v[1];
      if ((FAST && grp + 2 * t < 12) || p < L) store_h8(smem, p, ch0, v);
    }
  }
  if (mean_nx && tid < L) {
    stat_nx[tid] = st_m;
    stat_nx[MAXPOS + tid] = st_r;
  }
  if (part_o) {
    put_sum(sx0, sx1, 2);
    put_sum(sc0, sc1, 3);
  }
  __syncthreads();   // the new dc image and the next statistics are in place
}

template <int NT, int GRP, int NTHR, bool FAST>
__device__ __forceinline__ void train_bwd_body(const TrainBwdArgs& a, char* smem, int tid, int wave, int lane) {
  const int ct = wave & 3;
  constexpr int grp = GRP;
  const int L = a.L, nl = a.n_layers;
  const long b = blockIdx.x;
  const long M = (long)a.B * L;
  const float* vecs = (const float*)(smem + OFF_VECS);
  const float* tbl = (const float*)(smem + OFF_TB);
  auto dil_of = [&](int layer) { return (int)((a.dilpk[layer >> 3] >> (8 * (layer & 7))) & 255ull); };
  auto reach_of = [&](int layer) {
    const int d = dil_of(layer);
    if (FAST) return d == 64 ? (GRP == 1 ? 2 : 3) : 4;
    const int m = (d > 0) ? (L - 1) / d : 4;
    return m < 4 ? m : 4;
  };
  auto slot_of = [&](int i) { return (long)(a.slots == 2 ? (i & 1) : i); };
  const char* wuni = a.wfrag_t + ct * (2 * SUB_BYTES);
  const unsigned loff = (unsigned)lane * 16u;
  const char* wlane = wuni + loff;
  const int head0_tap = 9 * nl;

  for (int i = tid; i < (HPAD + ZROW + 1) * ROWB / 16; i += NTHR) ((f32x4*)smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < (nl + 1) * 3 * 128 / 4; i += NTHR) ((f32x4*)(smem + OFF_VECS))[i] = ((const f32x4*)a.vecs)[i];
  for (int i = tid; i < nl * 128 / 4; i += NTHR) ((f32x4*)(smem + OFF_TB))[i] = ((const f32x4*)a.tbias)[i];
  bf16v8 ring[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
    ring[f] = *(const bf16v8*)(wlane + (long)head0_tap * TAP_BYTES + (f >> 1) * KSTEP_BYTES + (f & 1) * SUB_BYTES);
  __syncthreads();   // the image is zero before the rows land in it
  {   // final_conv.0's gated output gradient of this sequence -> the image (whole 256-byte rows, 16-byte pieces)
    const char* src = a.dcs + (slot_of(nl) * M + b * L) * 256;
    for (int i = tid; i < L * 16; i += NTHR)
      *(f32x4*)(smem + OFF_HB + (i >> 4) * ROWB + 16 * (i & 15)) = *(const f32x4*)(src + (size_t)i * 16);
    // (statistics of the top residual layer's LayerNorm, for the first epilogue_lnb<LN>; two buffers, alternating by layer
    // parity: a layer's pass 2 reads its own while the next layer's are written)
    if (tid < L) {
      float* st0 = (float*)(smem + OFF_STAT) + ((nl - 1) & 1) * 2 * MAXPOS;
      st0[tid] = a.means[(long)(nl - 1) * M + b * L + tid];
      st0[MAXPOS + tid] = a.rstds[(long)(nl - 1) * M + b * L + tid];
    }
  }
  __syncthreads();

  f32x4 gr[NT][2], acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    gr[t][0] = gr[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int lo0 = opaque(lane);
  const int pn0 = (lo0 & 15) ^ ((lo0 & 8) >> 1);
  const int chl = 32 * ct + 4 * (lo0 >> 4);
  float* pslab = a.partials ? a.partials + (2 * b + grp) * a.part_stride : nullptr;
  const int wv = ct + 4 * grp;

  // ---- final_conv.0's input gradient (1x1): dL/dfeat_n, then the gate of the top residual layer -----------------------
  {
    const int m_top = reach_of(nl - 1);
    const long after = 9 * (nl - 1) + 4 - m_top;
    if constexpr (FAST) {
      const char* hcur = pad_base(smem, lane);
      bf16v8 cur[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(hcur + RI(GRP, t, 0));
      mma_tap_pad<NT, GRP, false>(hcur, hcur, wuni + after * TAP_BYTES, loff, ring, cur, acc);
    } else {
      const char* bp[NT];
      bf16v8 cur[NT];
      tap_begin<NT>(smem, 0, L, lane, grp, bp, cur);
      mma_tap<NT, false>(smem, wlane + after * TAP_BYTES, ring, 0, L, lane, grp, bp, cur, acc);
    }
    epilogue_lnb<NT, false, FAST>(smem, gr, acc, nullptr, false, nullptr, nullptr, L, lane, ct, grp,
                                  a.maskp + ((long)(nl - 1) * a.B + b) * DRK_MASKP_SEQ + wv * 64, pslab ? pslab + 512 * nl : nullptr, nullptr,
                                  nullptr, tid, nullptr, nullptr, nullptr, ring, wlane + after * TAP_BYTES);
  }

  int m_next = reach_of(nl - 1);
  for (int layer = nl - 1; layer >= 0; --layer) {
    const int dil = dil_of(layer);
    const int m = m_next;
    m_next = layer ? reach_of(layer - 1) : 0;
    const long after = layer ? 9 * (layer - 1) + 4 - m_next : 9 * layer + 4 - m;      // (layer 0: the refill is unused)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* wn = (FAST ? wuni : wlane) + (long)(9 * layer + 4 - m + 1) * TAP_BYTES;
    const char* wafter = (FAST ? wuni : wlane) + after * TAP_BYTES;
    if constexpr (FAST) {
      if (dil == 64) {
        static_layer<NT, GRP, 64, (GRP == 1 ? 2 : 3)>(smem, wn, wafter, loff, ring, lane, acc);
      } else if (dil == 16) {
        static_layer<NT, GRP, 16, 4>(smem, wn, wafter, loff, ring, lane, acc);
      } else {
        const char* hcur = pad_base(smem, lane) - 4 * dil * ROWB;
        const int hstep = dil * ROWB;
        bf16v8 cur[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(hcur + RI(GRP, t, 0));
        for (int k = 0; k < 8; k += 2) {
          TAP_PRIO(GRP == 0);
          mma_tap_pad<NT, GRP, true>(hcur, hcur + hstep, wn, loff, ring, cur, acc);
          hcur += hstep;
          wn += TAP_BYTES;
          TAP_PRIO(GRP != 0);
          mma_tap_pad<NT, GRP, true>(hcur, hcur + hstep, wn, loff, ring, cur, acc);
          hcur += hstep;
          wn += TAP_BYTES;
        }
        TAP_PRIO(GRP == 0);
        mma_tap_pad<NT, GRP, false>(hcur, hcur, wafter, loff, ring, cur, acc);
      }
    } else {
      const char* bp[NT];
      bf16v8 cur[NT];
      int off = -m * dil;
      tap_begin<NT>(smem, off, L, lane, grp, bp, cur);
      for (int k = 0; k < 2 * m; ++k) {
        off += dil;
        mma_tap<NT, true>(smem, wn, ring, off, L, lane, grp, bp, cur, acc);
        wn += TAP_BYTES;
      }
      mma_tap<NT, false>(smem, wafter, ring, 0, L, lane, grp, bp, cur, acc);
    }
    TAP_PRIO(false);
    epilogue_lnb<NT, true, FAST>(smem, gr, acc,
                                 layer ? a.featn + ((long)(layer - 1) * a.B + b) * DRK_FEATN_SEQ + wv * (14 * 256) : a.feat0 + b * L * 128,
                                 layer == 0, vecs + (3 * layer + 1) * 128, tbl + 128 * layer, L, lane,
                                 ct, grp, layer ? a.maskp + ((long)(layer - 1) * a.B + b) * DRK_MASKP_SEQ + wv * 64 : nullptr,
                                 pslab ? pslab + 512 * layer : nullptr, layer ? a.means + (long)(layer - 1) * M + b * L : nullptr,
                                 layer ? a.rstds + (long)(layer - 1) * M + b * L : nullptr, tid,
                                 a.slots != 2 ? a.dcs + (slot_of(layer) * M + b * L) * 256 : nullptr,
                                 (const float*)(smem + OFF_STAT) + (layer & 1) * 2 * MAXPOS,
                                 (float*)(smem + OFF_STAT) + ((layer + 1) & 1) * 2 * MAXPOS, ring, wlane + after * TAP_BYTES);
  }
  // ---- dL/d(stem pre-activation) = dL/dfeat_0 * (feat_0 > 0) -----------------------------------------------------
  {
    const float* f0 = a.feat0 + b * L * 128;
    float* o = a.dpre + b * L * 128;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int p = 16 * (grp + 2 * t) + pn0;
      if (p < L) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const f32x4 f = *(const f32x4*)(f0 + (size_t)(p * 128 + chl + 16 * s));
          const f32x4 gq = gr[t][s];
          *(f32x4*)(o + (size_t)(p * 128 + chl + 16 * s)) =
              f32x4{f[0] > 0.f ? gq[0] : 0.f, f[1] > 0.f ? gq[1] : 0.f, f[2] > 0.f ? gq[2] : 0.f, f[3] > 0.f ? gq[3] : 0.f};
        }
      }
    }
  }
}

template <bool FAST>
__global__ void __launch_bounds__(512, 2) k_train_bwd_bf16(const TrainBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4)
    train_bwd_body<7, 0, 512, FAST>(a, smem, tid, wave, lane);
  else if (FAST || a.L <= 208)
    train_bwd_body<6, 1, 512, FAST>(a, smem, tid, wave, lane);
  else
    train_bwd_body<7, 1, 512, false>(a, smem, tid, wave, lane);
}

// Host-side sizes of the lane-order blocks.
extern "C" long drk_train_fused_featn_floats(int B, int n_layers) { return (B <= 0 || n_layers <= 0) ? 0 : (long)n_layers * B * DRK_FEATN_SEQ; }
extern "C" long drk_train_fused_maskp_bytes(int B, int n_layers) { return (B <= 0 || n_layers <= 0) ? 0 : (long)n_layers * B * DRK_MASKP_SEQ * 8; }

static bool fused_shape_ok(int L, int n_layers) { return L <= MAXPOS && n_layers >= 1 && n_layers <= MAX_LAYERS; }

template <typename Args>
static int fused_dilations(Args& a, int L, int n_layers, const int* dilations, bool* fast, const char* who) {
  for (int i = 0; i < 4; ++i) a.dilpk[i] = 0;
  *fast = (L > 192 && L <= 208);
  for (int i = 0; i < n_layers; ++i) {
    if (dilations[i] < 0 || dilations[i] > 255) return drk_set_error_msg(DRK_EINVAL, who);
    a.dilpk[i >> 3] |= (unsigned long long)dilations[i] << (8 * (i & 7));
    *fast = *fast && ((dilations[i] >= 1 && dilations[i] <= 4) || dilations[i] == 16 || dilations[i] == 64);
  }
  return DRK_OK;
}

// once per (device, kernel): lets the launch ask for more than 64 KB of LDS.  Not an allocation.
static int fused_lds_attr(const void* kern, std::atomic<unsigned>* flags, const char* who) {
  int devid = 0;
  hipError_t e = hipGetDevice(&devid);
  if (e != hipSuccess) return drk_set_error(e, who);
  if (devid < 0 || devid >= DRK_MAX_DEVICES || !flags[devid].load(std::memory_order_acquire)) {
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return drk_set_error(e, who);
    if (devid >= 0 && devid < DRK_MAX_DEVICES) flags[devid].store(1u, std::memory_order_release);
  }
  return DRK_OK;
}

// The residual layers and final_conv.0 of one differentiable forward in one launch (see k_train_fwd_bf16): L <= 224 and
// <= 20 layers (DRK_EINVAL otherwise: drk_denoiser_fwd_bf16 then takes the per-layer launches).
extern "C" int drk_train_fwd_fused_bf16(const void* wstream, const float* vecs, const float* tbias, const float* feat0,
                                        float* featn, void* hs, void* maskp, float* means, float* rstds, float* y, int B, int L,
                                        int n_layers, const int* dilations, void* stream) {
  if (B < 0 || L <= 0) return drk_set_error_msg(DRK_EINVAL, "drk_train_fwd_fused_bf16: bad shape");
  if (B == 0) return DRK_OK;
  if (!fused_shape_ok(L, n_layers)) return drk_set_error_msg(DRK_EINVAL, "drk_train_fwd_fused_bf16: L <= 224 and 1..20 layers");
  if (!wstream || !vecs || !tbias || !feat0 || !featn || !hs || !maskp || !means || !rstds || !y || !dilations)
    return drk_set_error_msg(DRK_EINVAL, "drk_train_fwd_fused_bf16: null pointer");
  if (((uintptr_t)featn | (uintptr_t)maskp | (uintptr_t)hs | (uintptr_t)feat0 | (uintptr_t)y) & 15)
    return drk_set_error_msg(DRK_EINVAL, "drk_train_fwd_fused_bf16: 16-byte aligned blocks");
  TrainArgs a;
  a.wfrag = (const char*)wstream; a.vecs = vecs; a.tbias = tbias; a.feat0 = feat0; a.featn = featn; a.hs = (char*)hs;
  a.maskp = (unsigned long long*)maskp; a.means = means; a.rstds = rstds; a.y = y; a.B = B; a.L = L; a.n_layers = n_layers;
  bool fast;
  int rc = fused_dilations(a, L, n_layers, dilations, &fast, "drk_train_fwd_fused_bf16: dilation out of range");
  if (rc != DRK_OK) return rc;
  static std::atomic<unsigned> lds_set[2][DRK_MAX_DEVICES];
  rc = fused_lds_attr(fast ? (const void*)k_train_fwd_bf16<true> : (const void*)k_train_fwd_bf16<false>, lds_set[fast],
                      "drk_train_fwd_fused_bf16(LDS size)");
  if (rc != DRK_OK) return rc;
  if (fast)
    hipLaunchKernelGGL(k_train_fwd_bf16<true>, dim3(B), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_train_fwd_bf16<false>, dim3(B), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

// The backward of the residual layers and final_conv.0 in one launch (see k_train_bwd_bf16).  dcs: slot n_layers (mod
// slots) holds drk_head_bwd_bf16's output on entry; slots = n_layers + 1 keeps every layer's dc, 2 ping-pongs.
// partials: 2 B slabs of part_stride floats (two per sequence), slot i = [4][128] at 512 i as drk_layer_bwd_bf16 leaves
// them, i = n_layers for final_conv.0's launch; NULL: not wanted.  dpre: dL/d(stem pre-activation), fp32 [B*L][128].
extern "C" int drk_train_bwd_fused_bf16(const void* wstream_t, const float* vecs, const float* tbias, const float* feat0,
                                        const float* featn, const void* maskp, const float* means, const float* rstds, void* dcs,
                                        int dcs_slots, float* dpre, float* partials, long part_stride, int B, int L,
                                        int n_layers, const int* dilations, void* stream) {
  if (B < 0 || L <= 0) return drk_set_error_msg(DRK_EINVAL, "drk_train_bwd_fused_bf16: bad shape");
  if (B == 0) return DRK_OK;
  if (!fused_shape_ok(L, n_layers)) return drk_set_error_msg(DRK_EINVAL, "drk_train_bwd_fused_bf16: L <= 224 and 1..20 layers");
  if (!wstream_t || !vecs || !tbias || !feat0 || !featn || !maskp || !means || !rstds || !dcs || !dpre || !dilations)
    return drk_set_error_msg(DRK_EINVAL, "drk_train_bwd_fused_bf16: null pointer");
  if (dcs_slots != n_layers + 1 && dcs_slots != 2)
    return drk_set_error_msg(DRK_EINVAL, "drk_train_bwd_fused_bf16: dcs_slots is n_layers + 1 or 2");
  if (partials && (part_stride < 512L * (n_layers + 1) || (part_stride & 3) || ((uintptr_t)partials & 15)))
    return drk_set_error_msg(DRK_EINVAL, "drk_train_bwd_fused_bf16: partials: 16-byte aligned slabs of >= 512 (n_layers + 1) floats");
  if (((uintptr_t)featn | (uintptr_t)maskp | (uintptr_t)dcs | (uintptr_t)feat0 | (uintptr_t)dpre) & 15)
    return drk_set_error_msg(DRK_EINVAL, "drk_train_bwd_fused_bf16: 16-byte aligned blocks");
  TrainBwdArgs a;
  a.wfrag_t = (const char*)wstream_t; a.vecs = vecs; a.tbias = tbias; a.feat0 = feat0; a.featn = featn;
  a.maskp = (const unsigned long long*)maskp; a.means = means; a.rstds = rstds; a.dcs = (char*)dcs; a.slots = dcs_slots;
  a.dpre = dpre; a.partials = partials; a.part_stride = part_stride; a.B = B; a.L = L; a.n_layers = n_layers;
  bool fast;
  int rc = fused_dilations(a, L, n_layers, dilations, &fast, "drk_train_bwd_fused_bf16: dilation out of range");
  if (rc != DRK_OK) return rc;
  static std::atomic<unsigned> lds_set[2][DRK_MAX_DEVICES];
  rc = fused_lds_attr(fast ? (const void*)k_train_bwd_bf16<true> : (const void*)k_train_bwd_bf16<false>, lds_set[fast],
                      "drk_train_bwd_fused_bf16(LDS size)");
  if (rc != DRK_OK) return rc;
  if (fast)
    hipLaunchKernelGGL(k_train_bwd_bf16<true>, dim3(B), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_train_bwd_bf16<false>, dim3(B), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
