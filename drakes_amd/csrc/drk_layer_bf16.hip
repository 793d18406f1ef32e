// drk_layer_bf16 — ONE residual layer of the CNN denoiser (models/dnaconv.py:88-97) per launch, bf16 MFMA, for shapes
// the fused sampler cannot hold in LDS (L > 224: BASELINE.json configs[4], 1024-bp sequences) and for batched
// forwards:
//     feat' = feat + relu(conv_dilated(h) + bias)                      (fp32, global, in place allowed)
//     h'    = bf16( LayerNorm(feat' + tbias') * gamma' + beta' )       (the NEXT layer's input), or bf16(feat')
// h is the bf16 LayerNorm output of the previous launch ([rows][128], rows = B*L flattened).  A workgroup owns 128
// consecutive rows x all 128 output channels, so the LayerNorm of its rows completes inside it.
//   * MFMA v_mfma_f32_16x16x32_bf16, A = weights [16 co][32 ci] straight from L2 in fragment order (the SAME packed
//     stream and k-slot <-> channel map as drk_sample_bf16, fastpath.py::pack_stream), 8-deep register ring;
//   * B = activations, parked in LDS with 272-byte rows (conflict-free ds_read_b128 with the column permutation of
//     drk_sample_bf16).  Dilation <= 16: the 128 rows and their halo (up to 64 rows either side) are fetched ONCE as
//     whole 256-byte rows, a tap is a row offset, a lane whose shifted position leaves [0, L) reads an all-zero row
//     ('same' padding and the sequence boundary in one test), no barrier inside the tap loop, and the fp32 residual is
//     fetched ahead of the loop.  Dilation 64: each tap's 128 shifted rows are fetched separately, double-buffered,
//     the next tap's rows in flight in registers under this tap's 64 MFMAs per wave;
//   * epilogue: bias + ReLU + residual (feat read and written as 16-byte pieces), LayerNorm statistics by
//     v_permlane32_swap / v_permlane16_swap + four partials through LDS, h' staged in LDS and written out as rows.
// HBM per row and layer: 256 B (h) + 512 B + 512 B (fp32 residual in/out) + 256 B (h') = 1536 B against 295 kFLOP =
// 192 FLOP/B, below the machine balance (312 FLOP/B): HBM-bound once the tensors outgrow the 256 MB Infinity Cache.
#include "drk_common.h"
#include "../../include/drakes_hip.h"

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LT_ROWS 128              // rows per workgroup = 8 position tiles of 16
#define LT_NT 8
#define LT_ROWB 272              // LDS row: 128 bf16 + 16 B pad
#define LT_TAP_BYTES 32768       // [4 k-steps][4 ct][2 s][64 lanes][16 B]
#define LT_KSTEP_BYTES 8192
#define LT_SUB_BYTES 1024
#define LT_PART_STRIDE 12        // floats per row in the LayerNorm partial buffer (8 used)

// -DLT_STAMPS (tools/layer_stamps.py, never in the shipped library): thread 0 of every workgroup leaves wall-clock and
// shader-clock stamps of its phases in the mean / rstd outputs, which then hold no statistics.
#ifdef LT_STAMPS
#define LT_STAMP(i) do { if (threadIdx.x == 0) { ((unsigned long long*)a.mean_out)[blockIdx.x * 8 + (i)] = wall_clock64(); \
    ((unsigned long long*)a.rstd_out)[blockIdx.x * 8 + (i)] = clock64(); } } while (0)
#else
#define LT_STAMP(i) do { } while (0)
#endif

struct LayerArgs {
  const char* h;            // bf16 [M][128]
  const float* feat;        // fp32 [M][128] residual input (NULL: no residual, head conv)
  float* feat_out;          // fp32 [M][128]
  char* h_out;              // bf16 [M][128] (NULL: not needed)
  const char* wfrag;        // this layer's taps in fragment order
  const float* bias;        // [128] (NULL: none — the input-gradient use)
  unsigned char* mask_out;  // optional uint8 [M][128]: (pre-activation > 0), kept for the backward pass
  float* mean_out;          // optional [M]: statistics of the NEXT LayerNorm (ln = 1), kept for the backward pass
  float* rstd_out;
  int relu;                 // 0: feat_out = conv(h) (+ bias) (+ feat), no ReLU (the convolution's input gradient)
  const float* tbias;       // next LayerNorm's time bias [128] per batch row (stride tb_stride) or shared; NULL with ln = 0
  long tb_stride;
  const float* gamma;       // next LayerNorm's affine [128] (ln = 1)
  const float* beta;
  long M;
  int L, taps, dil, first_tap, ln;
  const float* head_w;      // drk_head_bf16 only: final_conv.2 [5][128] / [5]; the 128 -> 5 projection of the ReLU output
  const float* head_b;      //   is taken from the tile while it is on the chip, feat_out is not written
  float* logits;            //   float32 [M][5]
  // MODE 1 (drk_layer_bwd_bf16): h = dc of this layer, wfrag = its transposed stream, feat = the LayerNorm's INPUT, tbias /
  // gamma = that LayerNorm's (shared time bias only), mean_out / rstd_out are READ (the forward's statistics)
  const float* gr;          // dL/dfeat arriving through the residual branch, fp32 [M][128] (NULL: none)
  float* gout;              // dL/d(LayerNorm input) + gr, fp32 [M][128] (may be gr)
  const unsigned char* next_mask;   // ReLU gate of the layer below, uint8 [M][128] (NULL: no dc_next)
  char* dc_next;            // bf16 [M][128] = bf16(gout * next_mask)
  float* partials;          // [gridDim.x] slabs of part_stride floats; this launch fills [4][128] of its workgroup's slab:
  long part_stride;         //   column sums over its rows of dh * xhat, dh, dx, gated gout (NULL: not wanted)
};

// LDS: TROWS rows of 272 B (two 128-row tap buffers, or one halo tile), then the LayerNorm partials, whose first 272
// bytes are the all-zero row of HALO mode until the epilogue.

// HALO (128 + 2 * reach rows fit the tile: dilation <= 16 with nine taps): the workgroup's rows and their halo are
// fetched ONCE; a tap is a row offset, and a lane whose shifted position leaves [0, L) reads the all-zero row instead
// (sequence boundaries fall anywhere inside a tile).  No barrier inside the tap loop.
// !HALO (dilation 64): each tap's 128 shifted rows are fetched separately, double-buffered.
template <bool HALO, int TROWS, int WGS, bool PRE, int MODE = 0>
__global__ void __launch_bounds__(256, WGS) k_layer_bf16(const LayerArgs a) {
  constexpr int LT_TILE_BYTES = TROWS * LT_ROWB, LT_ZERO_OFF = LT_TILE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[LT_TILE_BYTES + LT_ROWS * LT_PART_STRIDE * 4];
  char* const tile0 = smem;
  float* const part = (float*)(smem + LT_TILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int pn = n ^ ((n & 8) >> 1);
  const int goff = 128 * (g & 1) + 64 * (g >> 1);
  const long r0 = (long)blockIdx.x * LT_ROWS;
  const int L = a.L;
  const unsigned p0 = (unsigned)(r0 % L);               // position of the tile's first row in its sequence

  // staging role: rows srow + 16 i (i < 8), 16-byte chunk scol
  const int srow = tid >> 4, scol = tid & 15;
  f32x4 stage[8];

  // the accumulators start at the conv bias (this lane's 8 channels): its load rides under the fetch of the rows instead
  // of a round trip at the head of the epilogue
  f32x4 acc[LT_NT][2];
  {
    const int chb = 32 * ct + 4 * g;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 cb0 = a.bias ? *(const f32x4*)(a.bias + chb) : z4, cb1 = a.bias ? *(const f32x4*)(a.bias + chb + 16) : z4;
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      acc[t][0] = cb0;
      acc[t][1] = cb1;
    }
  }
  LT_STAMP(0);
  const int center = a.taps / 2;
  const int last_tap = a.taps - 1 - a.first_tap;          // live taps: first_tap .. taps-1-first_tap (shift < L)
  const char* wlane = a.wfrag + ct * (2 * LT_SUB_BYTES) + lane * 16;
  bf16v8 ring[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
    ring[f] = *(const bf16v8*)(wlane + (long)a.first_tap * LT_TAP_BYTES + (f >> 1) * LT_KSTEP_BYTES + (f & 1) * LT_SUB_BYTES);

  // HALO has the registers to fetch the fp32 residual ahead of the tap loop instead of in the epilogue
  const int ch0 = 32 * ct + 4 * g;
  f32x4 res[PRE ? LT_NT : 1][2];
  if constexpr (HALO) {
    const int halo = center * a.dil;
    const int rows_total = LT_ROWS + 2 * halo;            // <= 256, checked by the entry point
    if (tid < LT_ROWB / 16) *(f32x4*)(smem + LT_ZERO_OFF + 16 * tid) = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < rows_total; c += LT_ROWS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int lr = c + srow + 16 * i;
        const long row = r0 - halo + lr;
        stage[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lr < rows_total && row >= 0 && row < a.M) stage[i] = *(const f32x4*)(a.h + row * 256 + 16 * scol);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int lr = c + srow + 16 * i;
        if (lr < rows_total) *(f32x4*)(tile0 + lr * LT_ROWB + 16 * scol) = stage[i];
      }
    }
    __syncthreads();
    LT_STAMP(1);
    // the fp32 residual of this lane's outputs: in flight under the whole tap loop
#pragma unroll
    for (int t = 0; t < (PRE ? LT_NT : 0); ++t) {
      const long row = r0 + 16 * t + pn;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        res[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.feat && row < a.M) res[t][s] = *(const f32x4*)(a.feat + row * DRK_C + ch0 + 16 * s);
      }
    }
    int pos[LT_NT];
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) pos[t] = (int)((p0 + 16 * t + pn) % (unsigned)L);
    const int own = (halo + pn) * LT_ROWB + goff;         // this lane's row of tile 0, unshifted
    // (branch-free on purpose: as a ?: the compiler built eight exec-masked blocks per tap, and the LDS reads in flight
    // across them were all waited for — lgkmcnt(0) — at every tap's first MFMA)
    auto addr_of = [&](int t, int s) {
      const int in = -(int)((unsigned)(pos[t] + s) < (unsigned)L);          // all ones inside [0, L)
      const int za = LT_ZERO_OFF + goff;
      return za + (in & (own + (16 * t + s) * LT_ROWB - za));
    };
    int cura[LT_NT];
    bf16v8 cur[LT_NT];
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) cura[t] = addr_of(t, (a.first_tap - center) * a.dil);
    // the first tap's fragments, pinned in tile order (as the reads inside the loop are): the wait at the loop header covers
    // both ways into it, and with tile 0's read issued last here it became lgkmcnt(0) at EVERY tap's first MFMA
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): no scalar load (they return out of order) is pending at the loop header
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      cur[t] = *(const bf16v8*)(smem + cura[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int tap = a.first_tap; tap <= last_tap; ++tap) {
      const int ntap = tap < last_tap ? tap + 1 : tap;    // (after the last tap the refills are unused)
      const char* wnext = wlane + (long)ntap * LT_TAP_BYTES;
      int nxta[LT_NT];
#pragma unroll
      for (int t = 0; t < LT_NT; ++t) nxta[t] = addr_of(t, (ntap - center) * a.dil);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16v8 a0 = ring[2 * ks], a1 = ring[2 * ks + 1];
#pragma unroll
        for (int t = 0; t < LT_NT; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, cur[t], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, cur[t], acc[t][1], 0, 0, 0);
          cur[t] = ks + 1 < 4 ? *(const bf16v8*)(smem + cura[t] + 16 * (ks + 1)) : *(const bf16v8*)(smem + nxta[t]);
        }
        ring[2 * ks] = *(const bf16v8*)(wnext + ks * LT_KSTEP_BYTES);
        ring[2 * ks + 1] = *(const bf16v8*)(wnext + ks * LT_KSTEP_BYTES + LT_SUB_BYTES);
#pragma unroll
        for (int t = 0; t < LT_NT; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
#pragma unroll
      for (int t = 0; t < LT_NT; ++t) cura[t] = nxta[t];
    }
  } else {
    int spos[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) spos[i] = (int)((p0 + srow + 16 * i) % (unsigned)L);
    auto fetch = [&](int off) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long row = r0 + srow + 16 * i;
        const int p = spos[i] + off;
        stage[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < a.M && p >= 0 && p < L) stage[i] = *(const f32x4*)(a.h + (row + off) * 256 + 16 * scol);
      }
    };
    auto park = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *(f32x4*)(tile0 + buf * (LT_ROWS * LT_ROWB) + (srow + 16 * i) * LT_ROWB + 16 * scol) = stage[i];
    };
    fetch((a.first_tap - center) * a.dil);
    park(0);
    __syncthreads();
    for (int tap = a.first_tap; tap <= last_tap; ++tap) {
      const int buf = (tap - a.first_tap) & 1;
      const bool more = tap < last_tap;
      if (more) fetch((tap + 1 - center) * a.dil);           // in flight under this tap's MFMAs
      const char* wnext = wlane + (long)(more ? tap + 1 : tap) * LT_TAP_BYTES;   // (after the last tap the refill is unused)
      const char* base = tile0 + buf * (LT_ROWS * LT_ROWB) + pn * LT_ROWB + goff;
      bf16v8 cur[LT_NT];
#pragma unroll
      for (int t = 0; t < LT_NT; ++t) cur[t] = *(const bf16v8*)(base + 16 * t * LT_ROWB);
      __builtin_amdgcn_sched_group_barrier(0x100, LT_NT, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16v8 a0 = ring[2 * ks], a1 = ring[2 * ks + 1];
#pragma unroll
        for (int t = 0; t < LT_NT; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, cur[t], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, cur[t], acc[t][1], 0, 0, 0);
          if (ks + 1 < 4) cur[t] = *(const bf16v8*)(base + 16 * t * LT_ROWB + 16 * (ks + 1));
        }
        ring[2 * ks] = *(const bf16v8*)(wnext + ks * LT_KSTEP_BYTES);
        ring[2 * ks + 1] = *(const bf16v8*)(wnext + ks * LT_KSTEP_BYTES + LT_SUB_BYTES);
#pragma unroll
        for (int t = 0; t < LT_NT; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          if (ks + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
      if (more) {
        park(buf ^ 1);       // the other buffer: its readers finished before the previous barrier
        __syncthreads();
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------
  LT_STAMP(2);
  {
#pragma clang fp contract(fast)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE == 1) {
    // ---- LayerNorm backward + residual add + hand-over to the layer below (drk_train_f32.hip::k_ln_bwd's arithmetic on
    // the tile while it is on the chip).  acc = dh = dL/dh of this layer's LayerNorm output;  xhat = (feat + tbias - mean)
    // * rstd;  dyg = dh * gamma;  dx = rstd * (dyg - mean_c(dyg) - xhat * mean_c(dyg * xhat));  gout = dx + gr;
    // dc_next = bf16(gout * next_mask).  Column sums of dh * xhat (dgamma), dh (dbeta), dx (time bias), gated gout (the
    // lower layer's conv bias) leave as one [4][128] partial per workgroup, summed in order by the caller.
    const bool ln = a.ln;
    const f32x4 tb0 = ln ? *(const f32x4*)(a.tbias + ch0) : zero4, tb1 = ln ? *(const f32x4*)(a.tbias + ch0 + 16) : zero4;
    const f32x4 gm0 = ln ? *(const f32x4*)(a.gamma + ch0) : zero4, gm1 = ln ? *(const f32x4*)(a.gamma + ch0 + 16) : zero4;
    float mu[LT_NT], rs[LT_NT];
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      long row = r0 + 16 * t + pn;
      row = row < a.M ? row : a.M - 1;
      mu[t] = ln ? a.mean_out[row] : 0.f;
      rs[t] = ln ? a.rstd_out[row] : 0.f;
    }
    f32x4 xh[LT_NT][2];
    if constexpr (!PRE) {
#pragma unroll
      for (int t = 0; t < LT_NT; ++t) {
        long row = r0 + 16 * t + pn;
        row = row < a.M ? row : a.M - 1;
        xh[t][0] = ln ? *(const f32x4*)(a.feat + row * DRK_C + ch0) : zero4;
        xh[t][1] = ln ? *(const f32x4*)(a.feat + row * DRK_C + ch0 + 16) : zero4;
      }
    } else {
#pragma unroll
      for (int t = 0; t < LT_NT; ++t) {
        xh[t][0] = res[t][0];
        xh[t][1] = res[t][1];
      }
    }
    f32x4 sg0 = zero4, sg1 = zero4, sb0 = zero4, sb1 = zero4;
    __syncthreads();   // every wave is past its reads of the activation tiles (and of the all-zero row)
    if (ln) {
      float y[LT_NT];
#pragma unroll
      for (int t = 0; t < LT_NT; ++t) {
        const bool ok = r0 + 16 * t + pn < a.M;
        const f32x4 d0 = ok ? acc[t][0] : zero4, d1 = ok ? acc[t][1] : zero4;
        const f32x4 h0 = (xh[t][0] + tb0 - mu[t]) * rs[t], h1 = (xh[t][1] + tb1 - mu[t]) * rs[t];
        sg0 += d0 * h0;
        sg1 += d1 * h1;
        sb0 += d0;
        sb1 += d1;
        const f32x4 y0 = d0 * gm0, y1 = d1 * gm1;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s1 += y0[i] + y1[i];
          s2 += y0[i] * h0[i] + y1[i] * h1[i];
        }
        acc[t][0] = y0;
        acc[t][1] = y1;
        xh[t][0] = h0;
        xh[t][1] = h1;
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s2), false, false);
        y[t] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const int slot = (g >> 1) * 4 + ct;
#pragma unroll
      for (int t = 0; t < LT_NT; t += 2) {
        const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[t]), __float_as_uint(y[t + 1]), false, false);
        const int tile_i = (g & 1) ? t + 1 : t;
        part[(16 * tile_i + pn) * LT_PART_STRIDE + slot] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < LT_NT; ++t) {
        const bool ok = r0 + 16 * t + pn < a.M;
        acc[t][0] = ok ? acc[t][0] : zero4;
        acc[t][1] = ok ? acc[t][1] : zero4;
      }
    }
    // the residual-branch gradient and the lower layer's gate: requested ahead of the barrier
    f32x4 gv[LT_NT][2];
    unsigned mk[LT_NT][2];
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      const long row = r0 + 16 * t + pn;
      const bool ok = row < a.M;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        gv[t][s] = (a.gr && ok) ? *(const f32x4*)(a.gr + row * DRK_C + ch0 + 16 * s) : zero4;
        mk[t][s] = (a.next_mask && ok) ? *(const unsigned*)(a.next_mask + row * DRK_C + ch0 + 16 * s) : 0u;
      }
    }
    __syncthreads();
    f32x4 sx0 = zero4, sx1 = zero4, sc0 = zero4, sc1 = zero4;
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      const long row = r0 + 16 * t + pn;
      const bool ok = row < a.M;
      f32x4 dx0 = acc[t][0], dx1 = acc[t][1];
      if (ln) {
        const float* pp = part + (16 * t + pn) * LT_PART_STRIDE;
        const f32x4 q1 = *(const f32x4*)(pp), q2 = *(const f32x4*)(pp + 4);
        const float m1 = ((q1[0] + q1[1]) + (q1[2] + q1[3])) * (1.0f / 128.0f);
        const float m2 = ((q2[0] + q2[1]) + (q2[2] + q2[3])) * (1.0f / 128.0f);
        dx0 = (dx0 - m1 - xh[t][0] * m2) * rs[t];
        dx1 = (dx1 - m1 - xh[t][1] * m2) * rs[t];
      }
      sx0 += dx0;
      sx1 += dx1;
      const f32x4 o0 = dx0 + gv[t][0], o1 = dx1 + gv[t][1];
      if (ok && a.gout) {
        *(f32x4*)(a.gout + row * DRK_C + ch0) = o0;
        *(f32x4*)(a.gout + row * DRK_C + ch0 + 16) = o1;
      }
      if (a.dc_next) {
        const unsigned m0 = mk[t][0], m1b = mk[t][1];
        const f32x4 v0 = {(m0 & 0xffu) ? o0[0] : 0.f, (m0 & 0xff00u) ? o0[1] : 0.f, (m0 & 0xff0000u) ? o0[2] : 0.f, (m0 & 0xff000000u) ? o0[3] : 0.f};
        const f32x4 v1 = {(m1b & 0xffu) ? o1[0] : 0.f, (m1b & 0xff00u) ? o1[1] : 0.f, (m1b & 0xff0000u) ? o1[2] : 0.f, (m1b & 0xff000000u) ? o1[3] : 0.f};
        sc0 += v0;
        sc1 += v1;
        char* rowp = tile0 + (16 * t + pn) * LT_ROWB + 2 * ch0;
        bf16v4 pk;
        pk[0] = (__bf16)v0[0]; pk[1] = (__bf16)v0[1]; pk[2] = (__bf16)v0[2]; pk[3] = (__bf16)v0[3];
        *(bf16v4*)(rowp) = pk;
        pk[0] = (__bf16)v1[0]; pk[1] = (__bf16)v1[1]; pk[2] = (__bf16)v1[2]; pk[3] = (__bf16)v1[3];
        *(bf16v4*)(rowp + 32) = pk;
      }
    }
    if (a.dc_next) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long row = r0 + srow + 16 * i;
        if (row < a.M) *(f32x4*)(a.dc_next + row * 256 + 16 * scol) = *(const f32x4*)(tile0 + (srow + 16 * i) * LT_ROWB + 16 * scol);
      }
    }
    if (a.partials) {
      // sum over the 16 rows a lane group holds (lanes 16 g + n, n = 0..15: one DPP row), in a fixed order
      f32x4 q[8] = {sg0, sg1, sb0, sb1, sx0, sx1, sc0, sc1};
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = q[k][i];
          v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
          v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
          v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
          v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, false));   // row_mirror
          q[k][i] = v;
        }
      if (n == 0) {
        float* dst = a.partials + (long)blockIdx.x * a.part_stride + ch0;
#pragma unroll
        for (int k = 0; k < 8; ++k) *(f32x4*)(dst + (k >> 1) * DRK_C + 16 * (k & 1)) = q[k];
      }
    }
    return;
  }
  // a time bias shared by the batch (the sampler's case) is requested here, two barriers before its use
  const bool tb_shared = a.ln && a.h_out && a.tb_stride == 0;
  const f32x4 tbs0 = tb_shared ? *(const f32x4*)(a.tbias + ch0) : zero4, tbs1 = tb_shared ? *(const f32x4*)(a.tbias + ch0 + 16) : zero4;
  const bool aff = a.ln && a.h_out;          // the next LayerNorm's affine: requested here as well
  const f32x4 g0 = aff ? *(const f32x4*)(a.gamma + ch0) : zero4, g1 = aff ? *(const f32x4*)(a.gamma + ch0 + 16) : zero4;
  const f32x4 b0 = aff ? *(const f32x4*)(a.beta + ch0) : zero4, b1 = aff ? *(const f32x4*)(a.beta + ch0 + 16) : zero4;
  f32x4 x[LT_NT][2];
  f32x4 late[PRE ? 1 : LT_NT][2];
  if constexpr (!PRE) {   // every residual request before the first use and the first store (feat_out may be feat, so
                          // the compiler cannot move a load above a store by itself): one round trip, not sixteen
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      long row = r0 + 16 * t + pn;
      row = row < a.M ? row : a.M - 1;
      late[t][0] = a.feat ? *(const f32x4*)(a.feat + row * DRK_C + ch0) : zero4;
      late[t][1] = a.feat ? *(const f32x4*)(a.feat + row * DRK_C + ch0 + 16) : zero4;
    }
  }
#pragma unroll
  for (int t = 0; t < LT_NT; ++t) {
    const long row = r0 + 16 * t + pn;
    const bool ok = row < a.M;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 r = acc[t][s];
      if (a.relu) {
        if (a.mask_out && ok) {   // this lane's 4 channels of one row: 4 bytes; the 4 lane groups of a row make 16 contiguous bytes
          const unsigned m = (r[0] > 0.f ? 1u : 0u) | (r[1] > 0.f ? 0x100u : 0u) | (r[2] > 0.f ? 0x10000u : 0u) |
                             (r[3] > 0.f ? 0x1000000u : 0u);
          *(unsigned*)(a.mask_out + row * DRK_C + ch0 + 16 * s) = m;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = fmaxf(r[i], 0.f);
      }
      if constexpr (PRE) r += res[t][s];
      else r += late[t][s];
      if (ok && a.feat_out) *(f32x4*)(a.feat_out + row * DRK_C + ch0 + 16 * s) = r;
      x[t][s] = r;
    }
  }
  LT_STAMP(3);
  if (a.logits) {   // (wave-uniform) head: y tile -> LDS as fp32 rows, then k_head_out's arithmetic on it, bit for bit
#pragma clang fp contract(off)
    __syncthreads();   // every wave is past its reads of the activation tiles
    float* const stg = (float*)tile0;                       // [128 rows][132 floats]: 67 584 B of the 69 632-byte tile area
    float* const ws = (float*)(smem + LT_TILE_BYTES);       // [5][128]: the partials' area (6 144 B)
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      *(f32x4*)(stg + (16 * t + pn) * 132 + ch0) = x[t][0];
      *(f32x4*)(stg + (16 * t + pn) * 132 + ch0 + 16) = x[t][1];
    }
    for (int i = tid; i < DRK_V * DRK_C; i += 256) ws[i] = a.head_w[i];
    __syncthreads();
    const int s8 = tid & 7;
#pragma unroll 1
    for (int pass = 0; pass < LT_ROWS / 32; ++pass) {
      const int rowl = 32 * pass + (tid >> 3);
      const long row = r0 + rowl;
      float part[DRK_V] = {0.f, 0.f, 0.f, 0.f, 0.f};
      if (row < a.M) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ch = 4 * (s8 + 8 * c);
          const f32x4 av = *(const f32x4*)(stg + rowl * 132 + ch);
#pragma unroll
          for (int v = 0; v < DRK_V; ++v) {
            const f32x4 ww = *(const f32x4*)(ws + v * DRK_C + ch);
            part[v] += (av[0] * ww[0] + av[1] * ww[1]) + (av[2] * ww[2] + av[3] * ww[3]);
          }
        }
      }
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) {
        part[v] += __shfl_xor(part[v], 4);
        part[v] += __shfl_xor(part[v], 2);
        part[v] += __shfl_xor(part[v], 1);
      }
      if (row < a.M && s8 < DRK_V) {
        float o = part[0];
#pragma unroll
        for (int v = 1; v < DRK_V; ++v) o = (s8 == v) ? part[v] : o;
        a.logits[row * DRK_V + s8] = o + a.head_b[s8];
      }
    }
    return;
  }
  if (!a.h_out) return;
  __syncthreads();   // every wave is past its reads of the activation tiles: tile[0] becomes the h' staging area
  if (a.ln) {
    float y[LT_NT];
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      const long row = r0 + 16 * t + pn;
      const long brow = (row < a.M ? row : a.M - 1) / L;
      if (tb_shared) {
        x[t][0] += tbs0;
        x[t][1] += tbs1;
      } else {
        const float* tb = a.tbias + brow * a.tb_stride + ch0;
        x[t][0] += *(const f32x4*)(tb);
        x[t][1] += *(const f32x4*)(tb + 16);
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s1 += x[t][s][i];
          s2 += x[t][s][i] * x[t][s][i];
        }
      const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s2), false, false);
      y[t] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const int slot = (g >> 1) * 4 + ct;
#pragma unroll
    for (int t = 0; t < LT_NT; t += 2) {
      const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[t]), __float_as_uint(y[t + 1]), false, false);
      const int tile_i = (g & 1) ? t + 1 : t;
      part[(16 * tile_i + pn) * LT_PART_STRIDE + slot] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    __syncthreads();
    LT_STAMP(4);
#pragma unroll
    for (int t = 0; t < LT_NT; ++t) {
      const float* pp = part + (16 * t + pn) * LT_PART_STRIDE;
      const f32x4 q1 = *(const f32x4*)(pp), q2 = *(const f32x4*)(pp + 4);
      const float s1 = (q1[0] + q1[1]) + (q1[2] + q1[3]);
      const float s2 = (q2[0] + q2[1]) + (q2[2] + q2[3]);
      const float mean = s1 * (1.0f / 128.0f);
      const float var = fmaxf(s2 * (1.0f / 128.0f) - mean * mean, 0.0f);
      const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
      const float c = -mean * rstd;
#ifndef LT_STAMPS
      if (a.mean_out && ct == 0 && g == 0 && r0 + 16 * t + pn < a.M) {
        a.mean_out[r0 + 16 * t + pn] = mean;
        a.rstd_out[r0 + 16 * t + pn] = rstd;
      }
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[t][0][i] = (x[t][0][i] * rstd + c) * g0[i] + b0[i];
        x[t][1][i] = (x[t][1][i] * rstd + c) * g1[i] + b1[i];
      }
    }
  }
  // h' tile -> LDS (natural channel order), then out as whole rows
#pragma unroll
  for (int t = 0; t < LT_NT; ++t) {
    char* row = tile0 + (16 * t + pn) * LT_ROWB + 2 * ch0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16v4 pk;
      pk[0] = (__bf16)x[t][s][0]; pk[1] = (__bf16)x[t][s][1]; pk[2] = (__bf16)x[t][s][2]; pk[3] = (__bf16)x[t][s][3];
      *(bf16v4*)(row + 32 * s) = pk;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long row = r0 + srow + 16 * i;
    if (row < a.M) *(f32x4*)(a.h_out + row * 256 + 16 * scol) = *(const f32x4*)(tile0 + (srow + 16 * i) * LT_ROWB + 16 * scol);
  }
  LT_STAMP(5);
#ifdef LT_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  LT_STAMP(6);
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    ((unsigned long long*)a.mean_out)[blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  }
}

// LayerNorm(feat + tbias) * gamma + beta -> bf16 rows: the first layer's input (after the stem).  32 lanes per row.
__global__ void __launch_bounds__(256) k_ln_bf16(const float* __restrict__ feat, const float* __restrict__ tbias, long tb_stride,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 char* __restrict__ h, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                 long M, int L) {
  const int lr = threadIdx.x & 31;
  const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= M) return;
  f32x4 v = *(const f32x4*)(feat + row * DRK_C + 4 * lr);
  if (tbias) v += *(const f32x4*)(tbias + (row / L) * tb_stride + 4 * lr);
  float s1 = (v[0] + v[1]) + (v[2] + v[3]);
  float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  const float mean = s1 * (1.0f / 128.0f);
  const float var = fmaxf(s2 * (1.0f / 128.0f) - mean * mean, 0.0f);
  const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
  if (mean_out && lr == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
  const f32x4 gm = *(const f32x4*)(gamma + 4 * lr), be = *(const f32x4*)(beta + 4 * lr);
  bf16v4 pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk[i] = (__bf16)(((v[i] - mean) * rstd) * gm[i] + be[i]);
  *(bf16v4*)(h + row * 256 + 8 * lr) = pk;
}

// Stem (token input) and the first LayerNorm in one launch: k_stem_tok's arithmetic (drk_cnn_f32.hip: one LDS-resident
// weight column per tap, same bits as drk_stem_fwd) with a half-wave holding a whole row, so k_ln_bf16's statistics run on
// the row while it is still in registers — feat is written, not read back (268 MB less per forward at 262 144 rows).
#define SLN_ROWS 64
__global__ void __launch_bounds__(256) k_stem_ln_bf16(const int64_t* __restrict__ x, const float* __restrict__ w_t,
                                                      const float* __restrict__ bias, const float* __restrict__ tbias,
                                                      long tb_stride, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ feat,
                                                      char* __restrict__ h, float* __restrict__ mean_out,
                                                      float* __restrict__ rstd_out, long M, int L) {
  __shared__ __attribute__((aligned(16))) float wl[DRK_TAPS * DRK_V * DRK_C];   // [tap][v][co]
  __shared__ int toks[SLN_ROWS + 8];
  const long r0 = (long)blockIdx.x * SLN_ROWS;
  for (int i = threadIdx.x; i < DRK_TAPS * DRK_V * DRK_C / 4; i += 256) ((f32x4*)wl)[i] = ((const f32x4*)w_t)[i];
  if (threadIdx.x < SLN_ROWS + 8) {
    const long r = r0 - 4 + threadIdx.x;
    toks[threadIdx.x] = (r >= 0 && r < M) ? (int)x[r] : -1;
  }
  const int lr = threadIdx.x & 31, rs = threadIdx.x >> 5;
  const f32x4 b4 = *(const f32x4*)(bias + 4 * lr);
  const f32x4 gm = *(const f32x4*)(gamma + 4 * lr), be = *(const f32x4*)(beta + 4 * lr);
  __syncthreads();
#pragma unroll 2
  for (int k = 0; k < SLN_ROWS / 8; ++k) {
    const int rr = rs + 8 * k;
    const long row = r0 + rr;
    if (row >= M) break;
    const int p = (int)(row % L);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DRK_TAPS; ++t) {
      const int pp = p + t - 4;
      const int tk = toks[rr + t];
      if (pp >= 0 && pp < L && (unsigned)tk < (unsigned)DRK_V) acc += *(const f32x4*)(wl + (t * DRK_V + tk) * DRK_C + 4 * lr);
    }
    f32x4 v = acc + b4;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
    *(f32x4*)(feat + row * DRK_C + 4 * lr) = v;
    // ---- k_ln_bf16 on the row (same operations in the same order)
    if (tbias) v += *(const f32x4*)(tbias + (row / L) * tb_stride + 4 * lr);
    float s1 = (v[0] + v[1]) + (v[2] + v[3]);
    float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    const float mean = s1 * (1.0f / 128.0f);
    const float var = fmaxf(s2 * (1.0f / 128.0f) - mean * mean, 0.0f);
    const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
    if (mean_out && lr == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
    bf16v4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = (__bf16)(((v[i] - mean) * rstd) * gm[i] + be[i]);
    *(bf16v4*)(h + row * 256 + 8 * lr) = pk;
  }
}

#define LARGCHECK(cond, name) \
  if (!(cond)) return drk_set_error_msg(DRK_EINVAL, name ": bad arguments")

extern "C" int drk_layer_bf16(const void* h, const float* feat, float* feat_out, void* h_out, const void* wfrag,
                              const float* bias, const float* tbias, long tb_stride, const float* gamma,
                              const float* beta, int B, int L, int taps, int dilation, int ln, int relu,
                              unsigned char* relu_mask, float* mean, float* rstd, void* stream) {
  DRK_SHAPE_GUARD("drk_layer_bf16");
  LARGCHECK(h && feat_out && wfrag && taps >= 1 && (taps & 1) && dilation >= 0, "drk_layer_bf16");
  LARGCHECK(!ln || (h_out && tbias && gamma && beta), "drk_layer_bf16 (LayerNorm parameters)");
  LARGCHECK((!mean && !rstd) || (mean && rstd && ln), "drk_layer_bf16 (mean / rstd come with ln = 1, both or neither)");
  LARGCHECK(!relu_mask || relu, "drk_layer_bf16 (a ReLU mask needs relu = 1)");
  LARGCHECK((const void*)h != (const void*)h_out, "drk_layer_bf16 (h and h_out must be distinct buffers)");
  LayerArgs a;
  a.mask_out = relu_mask; a.mean_out = mean; a.rstd_out = rstd; a.relu = relu ? 1 : 0;
  a.head_w = nullptr; a.head_b = nullptr; a.logits = nullptr;
  a.gr = nullptr; a.gout = nullptr; a.next_mask = nullptr; a.dc_next = nullptr; a.partials = nullptr; a.part_stride = 0;
  a.h = (const char*)h; a.feat = feat; a.feat_out = feat_out; a.h_out = (char*)h_out; a.wfrag = (const char*)wfrag;
  a.bias = bias; a.tbias = tbias; a.tb_stride = tb_stride; a.gamma = gamma; a.beta = beta;
  a.M = (long)B * L; a.L = L; a.taps = taps; a.dil = dilation; a.ln = ln ? 1 : 0;
  // taps whose shift is >= L read only padding at every position: skipped (as in drk_sample_bf16)
  const int center = taps / 2;
  int reach = center;
  if (dilation > 0 && (L - 1) / dilation < reach) reach = (L - 1) / dilation;
  a.first_tap = center - reach;
  const dim3 grid((unsigned)((a.M + LT_ROWS - 1) / LT_ROWS));
  const int rows_total = LT_ROWS + 2 * center * dilation;
  // (a 160-row, three-workgroups-per-CU variant for dilation <= 4 measured the same: the launch is HBM-bound)
  if (rows_total <= 256)             // rows + halo fetched once, residual prefetched
    hipLaunchKernelGGL((k_layer_bf16<true, 256, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else                               // one fetch per tap
    hipLaunchKernelGGL((k_layer_bf16<false, 256, 2, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_layer_bwd_bf16(const void* dc, const void* wfrag_t, const float* feat, const float* tbias, const float* gamma,
                                  const float* mean, const float* rstd, const float* gr, float* gout,
                                  const unsigned char* next_mask, void* dc_next, float* partials, long part_stride, int B,
                                  int L, int taps, int dilation, int ln, void* stream) {
  DRK_SHAPE_GUARD("drk_layer_bwd_bf16");
  LARGCHECK(dc && wfrag_t && gout && taps >= 1 && (taps & 1) && dilation >= 0, "drk_layer_bwd_bf16");
  LARGCHECK(!ln || (feat && tbias && gamma && mean && rstd), "drk_layer_bwd_bf16 (ln = 1 needs feat, tbias, gamma, mean, rstd)");
  LARGCHECK(!next_mask == !dc_next, "drk_layer_bwd_bf16 (next_mask and dc_next come together)");
  LARGCHECK((const void*)dc != (const void*)dc_next, "drk_layer_bwd_bf16 (dc and dc_next must be distinct buffers)");
  LARGCHECK(!partials || (part_stride >= 4 * DRK_C && (part_stride & 3) == 0 && ((uintptr_t)partials & 15) == 0),
            "drk_layer_bwd_bf16 (partials: 16-byte aligned slabs of part_stride >= 512 floats)");
  LayerArgs a;
  a.mask_out = nullptr; a.mean_out = const_cast<float*>(mean); a.rstd_out = const_cast<float*>(rstd); a.relu = 0;
  a.head_w = nullptr; a.head_b = nullptr; a.logits = nullptr;
  a.h = (const char*)dc; a.feat = ln ? feat : nullptr; a.feat_out = nullptr; a.h_out = nullptr; a.wfrag = (const char*)wfrag_t;
  a.bias = nullptr; a.tbias = tbias; a.tb_stride = 0; a.gamma = gamma; a.beta = nullptr;
  a.M = (long)B * L; a.L = L; a.taps = taps; a.dil = dilation; a.ln = ln ? 1 : 0;
  a.gr = gr; a.gout = gout; a.next_mask = next_mask; a.dc_next = (char*)dc_next; a.partials = partials; a.part_stride = part_stride;
  const int center = taps / 2;
  int reach = center;
  if (dilation > 0 && (L - 1) / dilation < reach) reach = (L - 1) / dilation;
  a.first_tap = center - reach;
  const dim3 grid((unsigned)((a.M + LT_ROWS - 1) / LT_ROWS));
  if (LT_ROWS + 2 * center * dilation <= 256)
    hipLaunchKernelGGL((k_layer_bf16<true, 256, 1, true, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_layer_bf16<false, 256, 1, false, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_ln_fwd_bf16(const float* feat, const float* tbias, long tb_stride, const float* gamma, const float* beta,
                               void* h, float* mean, float* rstd, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_ln_fwd_bf16");
  LARGCHECK(feat && gamma && beta && h && (!mean == !rstd), "drk_ln_fwd_bf16");
  const long M = (long)B * L;
  hipLaunchKernelGGL(k_ln_bf16, dim3((unsigned)((M + 7) / 8)), dim3(256), 0, (hipStream_t)stream, feat, tbias, tb_stride, gamma,
                     beta, (char*)h, mean, rstd, M, L);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_stem_ln_bf16(const int64_t* x, const float* w_t, const float* bias, const float* tbias, long tb_stride,
                                const float* gamma, const float* beta, float* feat, void* h, float* mean, float* rstd, int B,
                                int L, void* stream) {
  DRK_SHAPE_GUARD("drk_stem_ln_bf16");
  LARGCHECK(x && w_t && bias && gamma && beta && feat && h && (!mean == !rstd), "drk_stem_ln_bf16");
  const long M = (long)B * L;
  hipLaunchKernelGGL(k_stem_ln_bf16, dim3((unsigned)((M + SLN_ROWS - 1) / SLN_ROWS)), dim3(256), 0, (hipStream_t)stream, x, w_t,
                     bias, tbias, tb_stride, gamma, beta, feat, (char*)h, mean, rstd, M, L);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

extern "C" int drk_head_bf16(const void* h, const void* wfrag, const float* bias, const float* head_w, const float* head_b,
                             float* logits, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_head_bf16");
  LARGCHECK(h && wfrag && head_w && head_b && logits, "drk_head_bf16");
  LayerArgs a;
  a.mask_out = nullptr; a.mean_out = nullptr; a.rstd_out = nullptr; a.relu = 1;
  a.h = (const char*)h; a.feat = nullptr; a.feat_out = nullptr; a.h_out = nullptr; a.wfrag = (const char*)wfrag;
  a.bias = bias; a.tbias = nullptr; a.tb_stride = 0; a.gamma = nullptr; a.beta = nullptr;
  a.M = (long)B * L; a.L = L; a.taps = 1; a.dil = 0; a.ln = 0; a.first_tap = 0;
  a.head_w = head_w; a.head_b = head_b; a.logits = logits;
  a.gr = nullptr; a.gout = nullptr; a.next_mask = nullptr; a.dc_next = nullptr; a.partials = nullptr; a.part_stride = 0;
  const dim3 grid((unsigned)((a.M + LT_ROWS - 1) / LT_ROWS));
  hipLaunchKernelGGL((k_layer_bf16<true, 256, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
