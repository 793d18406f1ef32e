// drk_sample_bf16 — the whole reverse-sampling loop of Diffusion._sample
// (diffusion_gosai_update.py:506-550) as ONE persistent launch on gfx950.
//
// One workgroup (8 waves, two per SIMD) owns one sequence for all num_steps+1 denoiser
// forwards; sequences never interact, so there is no inter-workgroup traffic.
//   * the fp32 residual stream lives in REGISTERS, in the MFMA accumulator layout, for the
//     whole forward; it never touches LDS or HBM;
//   * the LayerNorm output h (bf16) lives in LDS, [position][128 ch] with padded rows, and is the
//     B operand of every tap of the dilated convolution: a tap is a row offset into that
//     image (rows outside [0,L) are redirected to one all-zero row = 'same' zero padding);
//   * conv weights (bf16, 5.9 MB for the whole network) are the A operand and go straight from
//     L2 into registers: they are packed in global memory in MFMA-fragment order, so a wave's
//     fragment is one contiguous, fully coalesced 1-KB load, and each wave keeps a ring of 8
//     fragments in flight (one whole tap ahead).  Weights never pass through LDS, so the LDS
//     carries only activation reads and no barrier is needed inside a layer;
//     every CU streams the same bytes at about the same time, so the stream stays L2-resident;
//   * the MFMA is v_mfma_f32_32x32x16_bf16 with A = weights [32 co][16 ci] and
//     B = activations [16 ci][32 positions]: the accumulator then holds, per lane, ONE position
//     and 16 channels, which makes the LayerNorm reduction a local sum plus one half-wave
//     exchange plus four partials through LDS, and the h write-back four 8-byte stores;
//   * stem (one-hot x 45 weights), final_conv.0/.2, SUBS, posterior, Philox, Gumbel-argmax and
//     carry-over are fused in; per step only the tokens change, and they stay in LDS; biases,
//     LayerNorm affine terms and time biases of all layers are LDS-resident.
// HBM traffic per sequence per step: ~0 (tokens out once at the end).  MFMA-bound.
//
// Work split: output tile = 32 channels x 32 positions; 4 channel tiles x 7 position tiles = 28
// tiles per k-step.  Wave w owns channel tile (w & 3) and position tiles (w >> 2) + 2t: waves
// 0-3 take 4 tiles, waves 4-7 take 3, and waves w, w+4 share a SIMD -> 7 tiles per SIMD.
#include "drk_common.h"
#include "../../include/drakes_hip.h"

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define ROWB 272                 // bytes per h-image row: 128 bf16 + 16 B pad.  Row r, 16-B chunk c sits in bank slot
                                 // (17r + c) mod 16, so 16 consecutive rows read conflict-free at a fixed chunk and a
                                 // k-step is an IMMEDIATE offset (+32 B) from a per-lane row address: no address VALU
#define MAXPOS 224               // 7 position tiles
#define ZROW 224                 // the all-zero row of the h image
#define TAP_BYTES 32768          // one tap of one layer in the fragment stream: [8 k-steps][4 co tiles][64 lanes][16 B]
#define KSTEP_BYTES 4096
#define STEM_ROWB 128            // resident stem image: [128 co][64 k] bf16 (k = 5*tap + v, 45 used)
#define MAX_LAYERS 20
#define OFF_HB 0
#define OFF_STEM (OFF_HB + (ZROW + 1) * ROWB)       // 57600
#define OFF_HEAD2 (OFF_STEM + 128 * STEM_ROWB)      // [8 rows][128] bf16, rows 5..7 zero
#define OFF_PART (OFF_HEAD2 + 8 * 256)             // float [2][4][224]: per channel-tile sum, sum of squares
#define OFF_LOG (OFF_PART + 2 * 4 * MAXPOS * 4)     // float [224][8]
#define OFF_TOK (OFF_LOG + MAXPOS * 8 * 4)          // int [224]
#define OFF_VECS (OFF_TOK + MAXPOS * 4)             // float [n_layers+1][3][128]
#define OFF_TB (OFF_VECS + (MAX_LAYERS + 1) * 3 * 128 * 4)   // float [n_layers][128]
#define LDS_BYTES (OFF_TB + MAX_LAYERS * 128 * 4)   // 133760 of 163840

struct SampleArgs {
  const char* wfrag;        // conv weights in fragment order: n_layers*9 taps, then final_conv.0 (fastpath.py)
  const char* stem_w;       // [128][64] bf16
  const char* head2_w;      // [8][128] bf16, rows 5..7 zero
  const float* vecs;        // [n_layers+1] blocks of [3][128]: block 0 = stem bias, gamma_0, beta_0; block k+1 =
                            // conv bias_k, gamma_{k+1}, beta_{k+1}; the last block = conv bias_{n-1}, head0 bias, head2 bias
  const float* tbias;       // [n_tb][n_layers][128]
  long tb_fwd_stride;       // floats between consecutive forwards' rows (0: one row for all)
  const float* mc_table;    // [num_steps][2] = (move_chance_t, move_chance_s)
  const float* uniforms;    // optional [num_steps][B][L][5]
  const int64_t* x_init;    // optional [B][L]; else all-mask prior
  int64_t* x_out;           // [B][L]
  float* logits_out;        // optional [B][L][5]: raw logits of forward `dump_fwd`
  int dump_fwd;
  float* dbg_feat;          // optional [B][L][128]: residual stream of forward dump_fwd after layer dbg_layer (-1: stem)
  int dbg_layer;
  int B, L, num_steps, n_layers, noise_removal;
  unsigned long long dilpk[4];   // 8 bits per layer
  uint64_t seed, sample_offset;
};

// Returns `v` through an empty asm: everything derived from the result is recomputed where it is used instead of
// being hoisted out of the layer/step loops and kept (then spilled) for the whole kernel.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// ---- one tap (or a 1x1 conv) ----------------------------------------------------------------
// acc[t] += W_tap[co tile] . h[rows of tile t shifted by `off`], K = 128 channels = 8 k-steps.
// ring[ks] holds this tap's weight fragments (loaded one tap ago); as each is consumed the SAME ring
// slot is refilled from `wnext` (this lane's pointer into the next live tap), so 8 global loads
// (one whole tap, ~2000 cycles) are always in flight.  (A two-tap ring was tried: the second copy of the tap
// code it needs for static register indexing pushed the allocator into scratch spills, 0.6x.)  Activation fragments of k-step ks+1 are read
// from LDS while the MFMAs of k-step ks run.
template <int NT, int TS>
__device__ __forceinline__ void mma_tap(const char* smem, const char* __restrict__ wnext, bf16v8 (&ring)[8], int off,
                                        int L, int lane, int grp, f32x16 (&acc)[NT]) {
  const int lo = opaque(lane);
  const int lr = lo & 31, half = lo >> 5;
  const char* bp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int pp = 32 * (grp + TS * t) + lr + off;
    const int row = ((unsigned)pp < (unsigned)L) ? pp : ZROW;   // 'same' zero padding = the all-zero row
    bp[t] = smem + OFF_HB + row * ROWB + 16 * half;
  }
  bf16v8 cur[NT], nxt[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(bp[t]);
  __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);   // DS reads of k-step 0
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const bf16v8 a = ring[ks];
#ifdef DRK_EXP_NOB    // timing experiment: activation fragments read once per tap
    if (false) {
#else
    if (ks + 1 < 8) {
#endif
#pragma unroll
      for (int t = 0; t < NT; ++t) nxt[t] = *(const bf16v8*)(bp[t] + 32 * (ks + 1));
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, cur[t], acc[t], 0, 0, 0);
#ifndef DRK_EXP_NOA   // timing experiment: no weight refill
    ring[ks] = *(const bf16v8*)(wnext + ks * KSTEP_BYTES);
#endif
#if !defined(DRK_EXP_NOA) && !defined(DRK_EXP_NOB)
    // Pinned order per k-step: MFMA, one LDS read of the next k-step, MFMA, read, ... so that every 32-cycle MFMA
    // slot carries one read's issue (a burst of NT reads after the first MFMA would stall the in-order wave past the
    // MFMA it could have issued), each read has a full k-step to land, and the refill of this ring slot comes
    // AFTER its last reader, so that the MFMAs read the slot in place (no register copy).
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (ks + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#endif
#ifndef DRK_EXP_NOB
    if (ks + 1 < 8) {
#pragma unroll
      for (int t = 0; t < NT; ++t) cur[t] = nxt[t];
    }
#endif
  }
}

// 16 per-channel values of this lane from an LDS vector of 128 floats: channels 32ct + 8g + 4half + e <-> register 4g + e.
__device__ __forceinline__ void load_vec16(const float* v, int ct, int half, float (&o)[16]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 q = *(const f32x4*)(v + 32 * ct + 8 * g + 4 * half);
    o[4 * g + 0] = q[0]; o[4 * g + 1] = q[1]; o[4 * g + 2] = q[2]; o[4 * g + 3] = q[3];
  }
}

// Writes 16 channel values of position p as bf16 into the h image.
__device__ __forceinline__ void store_h16(char* smem, int p, int ct, int half, const float (&x)[16]) {
  char* row = smem + OFF_HB + p * ROWB + 8 * half;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16v4 pk;
    pk[0] = (__bf16)x[4 * g + 0]; pk[1] = (__bf16)x[4 * g + 1];
    pk[2] = (__bf16)x[4 * g + 2]; pk[3] = (__bf16)x[4 * g + 3];
    *(bf16v4*)(row + ((4 * ct + g) << 4)) = pk;
  }
}

// Residual update feat += relu(acc + bias) [or feat = relu(acc + bias) after the stem], then, if `do_ln`, the next
// layer's LayerNorm of (feat + tbias) over the 128 channels of each position -> h image (bf16).
// blk: vector block [3][128] in LDS (bias, gamma, beta); tbv_src: the LayerNorm's time bias [128] in LDS.
// A lane holds 16 of a position's 128 channels per tile: local sums, one half-wave exchange (v_permlane32_swap,
// which leaves the sum in the lower and the sum of squares in the upper half-wave), four partials through LDS.
// Two barriers inside when do_ln; the first also orders "all MFMA reads of the old h image" before its overwrite.
template <int NT, int TS, bool ASSIGN>
__device__ __forceinline__ void epilogue_ln(char* smem, f32x16 (&feat)[NT], const f32x16 (&acc)[NT], const float* blk,
                                            const float* tbv_src, bool do_ln, int L, int lane, int ct, int grp) {
#pragma clang fp contract(fast)
  const int lo = opaque(lane);
  const int lr = lo & 31, half = lo >> 5;
  float cb[16], tbv[16];
  load_vec16(blk, ct, half, cb);
  if (do_ln) load_vec16(tbv_src, ct, half, tbv);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float r = fmaxf(acc[t][i] + cb[i], 0.f);
      feat[t][i] = ASSIGN ? r : feat[t][i] + r;
    }
  if (!do_ln) return;
  float* part = (float*)(smem + OFF_PART);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float x = feat[t][i] + tbv[i];
      s1 += x;
      s2 += x * x;
    }
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s2), false, false);
    // lower half-wave: s1(lower) + s1(upper); upper half-wave: s2(lower) + s2(upper)
    part[(half * 4 + ct) * MAXPOS + 32 * (grp + TS * t) + lr] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  float gv[16], bv[16];
  load_vec16(blk + 128, ct, half, gv);
  load_vec16(blk + 256, ct, half, bv);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = 32 * (grp + TS * t) + lr;
    const float s1 = (part[0 * MAXPOS + p] + part[1 * MAXPOS + p]) + (part[2 * MAXPOS + p] + part[3 * MAXPOS + p]);
    const float s2 = (part[4 * MAXPOS + p] + part[5 * MAXPOS + p]) + (part[6 * MAXPOS + p] + part[7 * MAXPOS + p]);
    const float mean = s1 * (1.0f / 128.0f);
    const float var = fmaxf(s2 * (1.0f / 128.0f) - mean * mean, 0.0f);
    const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
    const float c = -mean * rstd;
    float hv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) hv[i] = ((feat[t][i] + tbv[i]) * rstd + c) * gv[i] + bv[i];
    if (p < L) store_h16(smem, p, ct, half, hv);
  }
  __syncthreads();
}

template <int NT, int TS>
__device__ __forceinline__ void dump_feat(float* __restrict__ dst, long b, int L, const f32x16 (&feat)[NT], int lr,
                                          int half, int ct, int grp) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = 32 * (grp + TS * t) + lr;
    if (p < L) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 q = {feat[t][4 * g], feat[t][4 * g + 1], feat[t][4 * g + 2], feat[t][4 * g + 3]};
        *(f32x4*)(dst + (b * L + p) * 128 + 32 * ct + 8 * g + 4 * half) = q;
      }
    }
  }
}

// Diagnostic build only (-DDRK_STAMPS, tools/exp.py): per-wave cycle shares of the phases, written to dbg_feat as
// uint64 [B][8 waves][4] = (mma, -, layer epilogue+LayerNorm, stem/head/step).  Never in the product.
#ifdef DRK_STAMPS
#define STAMP(var)                                         \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    var = __builtin_amdgcn_s_memtime();                    \
    __builtin_amdgcn_s_waitcnt(0xC07F);                    \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
#define ACCUM(slot, t0v)                                   \
  do {                                                     \
    unsigned long long t1__;                               \
    STAMP(t1__);                                           \
    tacc[slot] += t1__ - t0v;                              \
    t0v = t1__;                                            \
  } while (0)
#else
#define STAMP(var) do {} while (0)
#define ACCUM(slot, t0v) do {} while (0)
#endif

// The body, specialised on the number of position tiles this wave owns (4 for waves 0-3, 3 for
// waves 4-7).  All 7 position tiles are always computed; positions >= L are never stored.
template <int NT, int TS, int NTHR>
__device__ __forceinline__ void sample_body(const SampleArgs& a, char* smem, int tid, int wave, int lane) {
  const int lr = lane & 31, half = lane >> 5;
  const int ct = wave & 3, grp = wave >> 2;
  const int L = a.L;
  const int ntiles = (L + 31) >> 5;
  const long b = blockIdx.x;
  const uint64_t sample = a.sample_offset + (uint64_t)b;
  int* tok = (int*)(smem + OFF_TOK);
  float* logit = (float*)(smem + OFF_LOG);
  const float* vecs = (const float*)(smem + OFF_VECS);
  const float* tbl = (const float*)(smem + OFF_TB);

  // A tap whose shift is >= L reads only padding at every position and is skipped altogether.
  // Live taps of a layer with dilation d: 4-m .. 4+m, m = min(4, (L-1)/d).
  auto dil_of = [&](int layer) { return (int)((a.dilpk[layer >> 3] >> (8 * (layer & 7))) & 255ull); };
  auto reach_of = [&](int layer) {
    const int d = dil_of(layer);
    const int m = (d > 0) ? (L - 1) / d : 4;
    return m < 4 ? m : 4;
  };
  // this lane's pointer to k-step 0 of tap index `tapidx` (= 9*layer + tap; 9*n_layers = final_conv.0)
  const char* wlane = a.wfrag + ct * 1024 + lane * 16;
  const int head0_tap = 9 * a.n_layers;
  const int first_tap = 4 - reach_of(0);

  // ---- init: tokens, zero row, resident weights and vectors, the first tap's fragments --------
  for (int i = tid; i < MAXPOS; i += NTHR) tok[i] = (i < L) ? (a.x_init ? (int)a.x_init[b * L + i] : DRK_MASK) : DRK_MASK;
  if (tid < ROWB / 4) ((int*)(smem + OFF_HB + ZROW * ROWB))[tid] = 0;   // the zero row (its pad bytes too)
  for (int i = tid; i < 128 * STEM_ROWB / 16; i += NTHR) ((f32x4*)(smem + OFF_STEM))[i] = ((const f32x4*)a.stem_w)[i];
  if (tid < 8 * 256 / 16) ((f32x4*)(smem + OFF_HEAD2))[tid] = ((const f32x4*)a.head2_w)[tid];
  for (int i = tid; i < (a.n_layers + 1) * 3 * 128 / 4; i += NTHR) ((f32x4*)(smem + OFF_VECS))[i] = ((const f32x4*)a.vecs)[i];
  // the flat sequence of live taps: ... layer l taps 4-m..4+m, layer l+1 ..., final_conv.0, next forward's first tap ...
  auto next_tap = [&](int idx) {
    if (idx == head0_tap) return first_tap;
    const int layer = idx / 9, tap = idx - 9 * layer;
    if (tap < 4 + reach_of(layer)) return idx + 1;
    return (layer + 1 < a.n_layers) ? 9 * (layer + 1) + 4 - reach_of(layer + 1) : head0_tap;
  };
  int t_n1 = next_tap(first_tap);   // the tap after the current one: what the current tap refills the ring with
  bf16v8 ring[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) ring[ks] = *(const bf16v8*)(wlane + (long)first_tap * TAP_BYTES + ks * KSTEP_BYTES);
#define RUN_TAP(off_)                                                                                   \
  do {                                                                                                  \
    mma_tap<NT, TS>(smem, wlane + (long)t_n1 * TAP_BYTES, ring, (off_), L, lane, grp, acc);             \
    t_n1 = next_tap(t_n1);                                                                              \
  } while (0)

#ifdef DRK_STAMPS
  unsigned long long tacc[4] = {0, 0, 0, 0}, t0;
  STAMP(t0);
#endif
  const int n_fwd = a.num_steps + (a.noise_removal ? 1 : 0);
  for (int f = 0; f < n_fwd; ++f) {
    f32x16 feat[NT], acc[NT];
    if (f == 0 || a.tb_fwd_stride != 0) {   // this forward's time biases (all layers) -> LDS
      const float* tb = a.tbias + (long)f * a.tb_fwd_stride;
      for (int i = tid; i < a.n_layers * 128 / 4; i += NTHR) ((f32x4*)(smem + OFF_TB))[i] = ((const f32x4*)tb)[i];
    }

    // ---- stem: one-hot(x) [45 -> 48 of K] x W_stem (LDS-resident), as 3 MFMA k-steps ----------
    __syncthreads();   // tokens / resident tables written above are visible
    {
      const int arow = 32 * ct + lr;
      const char* abase = smem + OFF_STEM + arow * STEM_ROWB + 16 * half;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const int p = 32 * (grp + TS * t) + lr;
        int nb[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int q = p + k - 4;
          nb[k] = ((unsigned)q < (unsigned)L) ? tok[q] : 7;
        }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const bf16v8 av = *(const bf16v8*)(abase + 32 * ks);
          bf16v8 bv;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            // k = 16ks + 8half + j = 5*tap + v
            const int k0 = 16 * ks + j, k1 = k0 + 8;
            const int t0 = k0 / 5, v0 = k0 % 5, t1 = k1 / 5, v1 = k1 % 5;
            const bool hit0 = (k0 < 45) && (nb[t0 < 9 ? t0 : 0] == v0);
            const bool hit1 = (k1 < 45) && (nb[t1 < 9 ? t1 : 0] == v1);
            bv[j] = (__bf16)((half ? hit1 : hit0) ? 1.0f : 0.0f);
          }
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
    epilogue_ln<NT, TS, true>(smem, feat, acc, vecs, tbl, true, L, lane, ct, grp);
    if (a.dbg_feat && f == a.dump_fwd && a.dbg_layer == -1) dump_feat<NT, TS>(a.dbg_feat, b, L, feat, lr, half, ct, grp);
    ACCUM(3, t0);

    // ---- n_layers x (dilated conv + ReLU + residual, then the next layer's LayerNorm) --------
    for (int layer = 0; layer < a.n_layers; ++layer) {
      const int dil = dil_of(layer);
      const int m = reach_of(layer);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
      for (int tap = 4 - m; tap <= 4 + m; ++tap) RUN_TAP((tap - 4) * dil);
      ACCUM(0, t0);
      epilogue_ln<NT, TS, false>(smem, feat, acc, vecs + 384 * (layer + 1), tbl + 128 * (layer + 1), layer + 1 < a.n_layers,
                             L, lane, ct, grp);
      if (a.dbg_feat && f == a.dump_fwd && a.dbg_layer == layer) dump_feat<NT, TS>(a.dbg_feat, b, L, feat, lr, half, ct, grp);
      ACCUM(2, t0);
    }

    // ---- head: final_conv.0 (1x1) + ReLU, final_conv.2 (128 -> 5) ---------------------------
    __syncthreads();   // the last layer did no LayerNorm: every wave is past its reads of the h image
    {
      const int lo_h = opaque(lane);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p = 32 * (grp + TS * t) + (lo_h & 31);
        float hv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) hv[i] = feat[t][i];
        if (p < L) store_h16(smem, p, ct, lo_h >> 5, hv);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
      }
    }
    __syncthreads();
    // the taps after final_conv.0 are the first conv taps of the NEXT forward: their fragments arrive under the
    // head, the reverse step and the next stem
    RUN_TAP(0);
    __syncthreads();   // all reads of the feat image are done
    {
      const int lo2 = opaque(lane);
      float hb0[16];
      load_vec16(vecs + 384 * a.n_layers + 128, ct, lo2 >> 5, hb0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p = 32 * (grp + TS * t) + (lo2 & 31);
        float hv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) hv[i] = fmaxf(acc[t][i] + hb0[i], 0.f);
        if (p < L) store_h16(smem, p, ct, lo2 >> 5, hv);
      }
    }
    __syncthreads();
    {
      // 5 output rows (LDS-resident weights, rows 5..7 zero, lanes >= 8 re-read row 7); position tiles dealt round-robin to waves
      for (int pt = wave; pt < ntiles; pt += NTHR / 64) {
        f32x16 lg;
#pragma unroll
        for (int i = 0; i < 16; ++i) lg[i] = 0.f;
        const int ar = (lr < 7) ? lr : 7;
        const char* abase = smem + OFF_HEAD2 + ar * 256 + 16 * half;
        const int p = 32 * pt + lr;
        const int row = (p < L) ? p : ZROW;
        const char* bbase = smem + OFF_HB + row * ROWB + 16 * half;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const bf16v8 av = *(const bf16v8*)(abase + 32 * ks);
          const bf16v8 bv = *(const bf16v8*)(bbase + 32 * ks);
          lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, lg, 0, 0, 0);
        }
        const float* hb2 = vecs + 384 * a.n_layers + 256;
        if (half == 0) {
          logit[p * 8 + 0] = lg[0] + hb2[0];
          logit[p * 8 + 1] = lg[1] + hb2[1];
          logit[p * 8 + 2] = lg[2] + hb2[2];
          logit[p * 8 + 3] = lg[3] + hb2[3];
        } else {
          logit[p * 8 + 4] = lg[0] + hb2[4];
        }
      }
    }
    __syncthreads();

    // ---- reverse step on this sequence's positions ------------------------------------------
    if (tid < L) {
      const int p = tid;
      float lgv[DRK_V];
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) lgv[v] = logit[p * 8 + v];
      if (a.logits_out && f == a.dump_fwd) {
#pragma unroll
        for (int v = 0; v < DRK_V; ++v) a.logits_out[(b * L + p) * DRK_V + v] = lgv[v];
      }
      const int tk = tok[p];
      if (f < a.num_steps) {
        if (tk == DRK_MASK) {
          float lp[DRK_V], q[DRK_V], u[DRK_V];
          if (a.uniforms) {
#pragma unroll
            for (int v = 0; v < DRK_V; ++v) u[v] = a.uniforms[(((long)f * a.B + b) * L + p) * DRK_V + v];
          } else {
            drk_step_uniforms(a.seed, (uint32_t)f, sample, (uint32_t)p, u);
          }
          drk_subs_logp(lgv, DRK_MASK, lp);
          drk_posterior(lp, a.mc_table[2 * f], a.mc_table[2 * f + 1], q);
          tok[p] = drk_gumbel_argmax(q, u);
        }
      } else {  // noise removal: SUBS, argmax over the four nucleotides
        float lp[DRK_V];
        drk_subs_logp(lgv, tk, lp);
        int best = 0;
#pragma unroll
        for (int v = 1; v < DRK_V - 1; ++v)
          if (lp[v] > lp[best]) best = v;
        tok[p] = best;
      }
    }
    ACCUM(3, t0);
  }
  __syncthreads();
#ifdef DRK_STAMPS
  if (a.dbg_feat && lane == 0) {
    unsigned long long* o = (unsigned long long*)a.dbg_feat + (b * 8 + wave) * 4;   // [B][8][4] also in the 4-wave build
    o[0] = tacc[0]; o[1] = tacc[1]; o[2] = tacc[2]; o[3] = tacc[3];
  }
#endif
  if (tid < L) a.x_out[b * L + tid] = tok[tid];
#undef RUN_TAP
}

// Two builds of the same body.  8 waves: two per SIMD, wave w = channel tile (w&3) x position tiles (w>>2)+2t
// (4 or 3 tiles, 256 registers).  4 waves: one per SIMD, wave w = channel tile w x all 7 position tiles (512
// registers): every weight fragment is then fetched from L2 by exactly ONE wave per CU (32 KB per tap instead of
// 64), which is what bounds the 8-wave build (~30 B/clk/CU of L2->register bandwidth).
#ifndef DRK_WAVES
#define DRK_WAVES 8
#endif
#if DRK_WAVES == 8
__global__ void __launch_bounds__(512, 2) k_sample_bf16(const SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches, SGPR tile ids
  if (wave < 4)
    sample_body<4, 2, 512>(a, smem, tid, wave, lane);
  else
    sample_body<3, 2, 512>(a, smem, tid, wave, lane);
}
#else
__global__ void __launch_bounds__(256, 1) k_sample_bf16(const SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  sample_body<7, 1, 256>(a, smem, tid, wave, lane);
}
#endif

// ------------------------------------------------------------------------------------ C ABI
extern "C" int drk_sample_bf16(const void* wstream, const void* stem_w, const void* head2_w, const float* vecs, const float* tbias, long tb_fwd_stride,
                               const float* mc_table, const float* uniforms, const int64_t* x_init, int64_t* x_out,
                               float* logits_out, int dump_fwd, float* dbg_feat, int dbg_layer, int B, int L, int num_steps, int n_layers,
                               const int* dilations, int noise_removal, uint64_t seed, uint64_t sample_offset,
                               void* stream) {
  if (B < 0 || L <= 0) return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: bad shape");
  if (B == 0) return DRK_OK;
  if (L > MAXPOS) return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: L > 224 is not supported by the fused loop");
  if (n_layers < 1 || n_layers > MAX_LAYERS || num_steps < 0 || (num_steps == 0 && !noise_removal))
    return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: bad n_layers / num_steps");
  if (!wstream || !stem_w || !head2_w || !vecs || !tbias || !x_out || !dilations || (num_steps > 0 && !mc_table))
    return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: null pointer");
  SampleArgs a;
  a.wfrag = (const char*)wstream; a.stem_w = (const char*)stem_w; a.head2_w = (const char*)head2_w; a.vecs = vecs; a.tbias = tbias; a.tb_fwd_stride = tb_fwd_stride;
  a.mc_table = mc_table; a.uniforms = uniforms; a.x_init = x_init; a.x_out = x_out; a.logits_out = logits_out;
  a.dump_fwd = dump_fwd; a.dbg_feat = dbg_feat; a.dbg_layer = dbg_layer; a.B = B; a.L = L; a.num_steps = num_steps; a.n_layers = n_layers;
  a.noise_removal = noise_removal ? 1 : 0; a.seed = seed; a.sample_offset = sample_offset;
  for (int i = 0; i < 4; ++i) a.dilpk[i] = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (dilations[i] < 0 || dilations[i] > 255) return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: dilation out of range");
    a.dilpk[i >> 3] |= (unsigned long long)dilations[i] << (8 * (i & 7));
  }
  {  // per device, cheap, and not an allocation: lets the launch ask for more than 64 KB of LDS
    hipError_t e = hipFuncSetAttribute((const void*)k_sample_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return drk_set_error(e, "drk_sample_bf16(hipFuncSetAttribute)");
  }
  hipLaunchKernelGGL(k_sample_bf16, dim3(B), dim3(64 * DRK_WAVES), LDS_BYTES, (hipStream_t)stream, a);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
