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
//   * the MFMA is v_mfma_f32_16x16x32_bf16 with A = weights [16 co][32 ci] and
//     B = activations [32 ci][16 positions].  16-position tiles cover L=200 with 13 tiles (208
//     positions, 4 % padding; 32-position tiles needed 224, 12 %), and the chip holds a higher clock on
//     this shape than on 32x32x16 (MI355X_MICROARCH.md, DVFS item 7).  The accumulator holds, per lane,
//     ONE position and 4 channels per 16x16 tile, which makes the LayerNorm reduction a local sum,
//     two cross-lane swaps and four partials through LDS, and the h write-back 8-byte stores;
//   * stem (an MFMA over K = 9 taps x 8 one-hot slots, the one-hot built in registers), final_conv.0/.2, SUBS, posterior, Philox,
//     Gumbel-argmax and carry-over are fused in; per step only the tokens change, and they stay in LDS;
//     biases, LayerNorm affine terms and time biases of all layers are LDS-resident.
// HBM traffic per sequence per step: ~0 (tokens out once at the end).  MFMA-bound.
//
// Work split: a wave owns 32 output channels (two 16-channel MFMA row tiles, s = 0,1) and every second
// 16-position tile: wave w has channel tile ct = w & 3 and position tiles (w >> 2) + 2t.  Waves 0-3 take 7 tiles,
// waves 4-7 take 6 (7 when L > 208); waves w and w+4 share a SIMD -> 13 tiles x 2 row tiles per SIMD per k-step.
//
// Lane <-> data.  lane = 16 g + n.  B operand: lane reads 8 consecutive stored channels of ONE position; A operand:
// 8 consecutive k of ONE output channel; D: D[row 4g+i][col n], i = 0..3.
//   position of column n inside a tile: pn = n ^ ((n & 8) >> 1)   (columns 8-11 <-> positions 12-15)
//   k slot (k-step ks, lane group g, j) <-> input channel 64 (g&1) + 32 (g>>1) + 8 ks + j
// Both permutations exist only to make ds_read_b128 conflict-free: the hardware serves a b128 read in lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS table), i.e. 8 lanes of group g and 8 of
// g+1; with 272-byte rows the bank slot is (row + 16-byte chunk) mod 16, and the two halves land on disjoint
// slots for every tap shift exactly when g and g+1 sit 8 chunks apart and the two sets of rows are {0-3,8-11} and
// {4-7,12-15}.  The weight stream is packed with the same k order (fastpath.py), the h image stays in natural
// channel order.
#include "drk_fused_common.h"

struct SampleArgs {
  const char* wfrag;        // conv weights in fragment order: n_layers*9 taps, then final_conv.0 (fastpath.py)
  const char* stem_w;       // stem weights in fragment order, K = 8*tap + v padded to 96: [3][4][2][64][8] bf16 (fastpath.py)
  const char* head2_w;      // [8][128] bf16, rows 5..7 zero
  const float* vecs;        // [n_layers+1] blocks of [3][128]: block 0 = stem bias, gamma_0, beta_0; block k+1 =
                            // conv bias_k, gamma_{k+1}, beta_{k+1}; the last block = conv bias_{n-1}, head0 bias, head2 bias
  const float* tbias;       // [n_tb][n_layers][128]
  long tb_fwd_stride;       // floats between consecutive forwards' rows (0: one row for all)
  const float* mc_table;    // [num_steps][2] = (move_chance_t, move_chance_s)
  const float* uniforms;    // optional [num_steps][B][L][5]
  const int64_t* x_init;    // optional [B][L]; else all-mask prior
  int64_t* x_out;           // [B][L]
  signed char* traj_out;    // optional [n_fwd][B][L]: the state each forward starts from (what _ddpm_update records)
  float* logits_out;        // optional [B][L][5]: raw logits of forward `dump_fwd`
  int dump_fwd;
  float* dbg_feat;          // optional [B][L][128]: residual stream of forward dump_fwd after layer dbg_layer (-1: stem)
  int dbg_layer;
  int B, L, num_steps, n_layers, noise_removal;
  unsigned long long dilpk[4];   // 8 bits per layer
  uint64_t seed, sample_offset;
};

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

// The body, specialised on the number of position tiles this wave owns (7 for waves 0-3, 6 or 7 for
// waves 4-7).  Positions >= L inside a computed tile are never stored.
// Residual update feat += relu(acc + bias) [or feat = relu(acc + bias) after the stem], then, if `do_ln`, the next
// layer's LayerNorm of (feat + tbias) over the 128 channels of each position -> h image (bf16).
// blk: vector block [3][128] in LDS (bias, gamma, beta); tbv_src: the LayerNorm's time bias [128] in LDS.
// A lane holds 8 of a position's 128 channels per tile, the 4 lanes n, n+16, n+32, n+48 hold 32, the 4 waves of a
// position group all 128: local sums; v_permlane32_swap (lanes +-32: leaves the sum in the lower and the sum of
// squares in the upper half-wave); v_permlane16_swap on a PAIR of tiles (lanes +-16: row 0/1 = sum of tile a/b,
// row 2/3 = sum of squares of tile a/b), so every lane ends with one finished statistic; four partials through LDS.
// Two barriers inside when do_ln; the first also orders "all MFMA reads of the old h image" before its overwrite.
// FAST (L > 192): only tile 12 holds positions >= L, the store guard of every other tile folds away.
template <int NT, bool ASSIGN, bool FAST>
__device__ __forceinline__ void epilogue_ln(char* smem, f32x4 (&feat)[NT][2], const f32x4 (&acc)[NT][2], const float* blk,
                                            const float* tbv_src, bool do_ln, int L, int lane, int ct, int grp
#ifdef DRK_STAMPS
                                            , unsigned long long (&tacc)[8], unsigned long long& t0
#endif
                                            ) {
#pragma clang fp contract(fast)
  const int lo = opaque(lane);
  const int n = lo & 15, g = lo >> 4;
  const int pn = n ^ ((n & 8) >> 1);
  const int ch0 = 32 * ct + 4 * g;
  // Written on whole f32x4 values so that the adds and the fused multiply-adds come out packed (v_pk_add_f32 /
  // v_pk_fma_f32, two channels per instruction): this pass shares the SIMD's vector issue with the other wave's MFMAs.
  // ASSIGN (stem): acc holds the bare sum and the bias is added here (keeps the stem bit-equal to its reference);
  // otherwise the conv bias is already in acc (the accumulators START at the bias).
  f32x4 tbv[2];
  if (do_ln) load_vec8(tbv_src, ch0, tbv);
  // ReLU as a signed-integer max with 0 on the bit pattern (negative floats and -0.0 are negative integers): one
  // v_max_i32 per value; a float max straight on MFMA results costs a canonicalising second instruction.
  auto relu4 = [](f32x4 v) __attribute__((always_inline)) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 z = {0, 0, 0, 0};
    return __builtin_bit_cast(f32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, v), z));
  };
  if constexpr (ASSIGN) {
    f32x4 cb[2];
    load_vec8(blk, ch0, cb);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) feat[t][s] = relu4(acc[t][s] + cb[s]);
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) feat[t][s] += relu4(acc[t][s]);
  }
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
    // lower half-wave: s1(lower) + s1(upper); upper half-wave: s2(lower) + s2(upper)
    y[t] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const int stat_ct = (g >> 1) * 4 + ct;
#pragma unroll
  for (int t = 0; t < NT; t += 2) {
    const int tb = (t + 1 < NT) ? t + 1 : t;   // odd count: the last tile pairs with itself
    const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(y[t]), __float_as_uint(y[tb]), false, false);
    // rows 0,2 (g even): tile t; rows 1,3 (g odd): tile tb
    const int tile = (g & 1) ? tb : t;
    part[(16 * (grp + 2 * tile) + pn) * PART_STRIDE + stat_ct] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  f32x4 gv[2], bv[2];
  load_vec8(blk + 128, ch0, gv);
  load_vec8(blk + 256, ch0, bv);
  ACCUM(4, t0);   // (stamps build) pass 1 up to the first barrier
  __syncthreads();
  ACCUM(1, t0);   // the wait at the first barrier
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
    if ((FAST && grp + 2 * t < 12) || p < L) store_h8(smem, p, ch0, hv);
  }
  __syncthreads();
}

template <int NT>
__device__ __forceinline__ void dump_feat(float* __restrict__ dst, long b, int L, const f32x4 (&feat)[NT][2], int lane,
                                          int ct, int grp) {
  const int lo = opaque(lane);   // debug path: nothing of it may be hoisted into the sampling loop
  const int pn = (lo & 15) ^ ((lo & 8) >> 1);
  const int ch0 = 32 * ct + 4 * (lo >> 4);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = 16 * (grp + 2 * t) + pn;
    if (p < L) {
      *(f32x4*)(dst + (b * L + p) * 128 + ch0) = feat[t][0];
      *(f32x4*)(dst + (b * L + p) * 128 + ch0 + 16) = feat[t][1];
    }
  }
}

// FAST: 192 < L <= 208 and every dilation in {1..4, 16, 64} (checked by the entry point): every layer runs on
// immediate addressing (zero-padded image for dilation <= 4, static live ranges for 16 and 64); the generic tap loop is
// not even compiled in, which also keeps its address registers out of the allocation.
template <int NT, int GRP, int NTHR, bool FAST>
__device__ __forceinline__ void sample_body(const SampleArgs& a, char* smem, int tid, int wave, int lane) {
  const int ct = wave & 3;
  constexpr int grp = GRP;
  const int L = a.L;
  const int ntiles = (L + 15) >> 4;
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
    // (FAST, dilation 64: the taps shifted by +-192 have no live tile among the odd tiles, so waves 4-7 skip them
    // altogether, weight fetch included; waves 0-3 keep tile 12 / tile 0)
    if (FAST) return d == 64 ? (GRP == 1 ? 2 : 3) : 4;
    const int m = (d > 0) ? (L - 1) / d : 4;
    return m < 4 ? m : 4;
  };
  // this lane's pointer to fragment (k-step 0, row tile 0) of tap index `tapidx` (= 9*layer + tap; 9*n_layers = final_conv.0)
  const char* wuni = a.wfrag + ct * (2 * SUB_BYTES);   // wave-uniform part, and this lane's 16 bytes of a fragment
  const unsigned loff = (unsigned)lane * 16u;
  const char* wlane = wuni + loff;
  const int head0_tap = 9 * a.n_layers;
  const int first_tap = 4 - reach_of(0);

  // ---- init: tokens, zero row, resident weights and vectors, the first tap's fragments --------
  for (int i = tid; i < MAXPOS; i += NTHR) tok[i] = (i < L) ? (a.x_init ? (int)a.x_init[b * L + i] : DRK_MASK) : DRK_MASK;
  // the whole h image starts as zeros: its padding rows, rows L..223 and the zero row are never written afterwards
  for (int i = tid; i < (HPAD + ZROW + 1) * ROWB / 16; i += NTHR) ((f32x4*)smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < STEM_BYTES / 16; i += NTHR) ((f32x4*)(smem + OFF_STEM))[i] = ((const f32x4*)a.stem_w)[i];
  if (tid < 8 * 16) *(f32x4*)(smem + OFF_HEAD2 + (tid >> 4) * HEAD2_ROWB + (tid & 15) * 16) = ((const f32x4*)a.head2_w)[tid];
  for (int i = tid; i < (a.n_layers + 1) * 3 * 128 / 4; i += NTHR) ((f32x4*)(smem + OFF_VECS))[i] = ((const f32x4*)a.vecs)[i];
  // The ring always holds the weight fragments of the tap about to run and is refilled, k-step by k-step, with those
  // of the NEXT live tap in the flat order ... layer l taps 4-m..4+m, layer l+1 ..., final_conv.0, next forward ...
  bf16v8 ring[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
    ring[f] = *(const bf16v8*)(wlane + (long)first_tap * TAP_BYTES + (f >> 1) * KSTEP_BYTES + (f & 1) * SUB_BYTES);

#ifdef DRK_STAMPS
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0;
  STAMP(t0);
  const unsigned long long clk0 = t0, rt0 = __builtin_amdgcn_s_memrealtime();   // shader cycles vs the 100 MHz counter
#endif
  const int n_fwd = a.num_steps + (a.noise_removal ? 1 : 0);
  for (int f = 0; f < n_fwd; ++f) {
    f32x4 feat[NT][2], acc[NT][2];
    if (f == 0 || a.tb_fwd_stride != 0) {   // this forward's time biases (all layers) -> LDS
      const float* tb = a.tbias + (long)f * a.tb_fwd_stride;
      for (int i = tid; i < a.n_layers * 128 / 4; i += NTHR) ((f32x4*)(smem + OFF_TB))[i] = ((const f32x4*)tb)[i];
    }

    // ---- stem: Conv1d(5 -> 128, k = 9) of a one-hot input as an MFMA with K = 8*tap + v (72, padded to 96) --------
    // B operand = the one-hot itself, built in registers: in k-step ks lane group g holds the 8 slots of tap 4ks + g,
    // i.e. bf16 1.0 in slot tok[p + tap - 4] (nothing for a padding position or tap > 8); A = LDS-resident fragments.
    __syncthreads();   // tokens / resident tables written above are visible
    if (a.traj_out && tid < L) a.traj_out[((long)f * a.B + b) * L + tid] = (signed char)tok[tid];
    {
      const int lo_s = opaque(lane);
      const int g_s = lo_s >> 4;
      const int pn_s = (lo_s & 15) ^ ((lo_s & 8) >> 1);
      const char* sfrag = smem + OFF_STEM + ct * (2 * SUB_BYTES) + lo_s * 16;
      bf16v8 sw[3][2];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        sw[ks][0] = *(const bf16v8*)(sfrag + ks * KSTEP_BYTES);
        sw[ks][1] = *(const bf16v8*)(sfrag + ks * KSTEP_BYTES + SUB_BYTES);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p = 16 * (grp + 2 * t) + pn_s;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int tapi = 4 * ks + g_s;
          const int q = p + tapi - 4;
          const bool live = (tapi < DRK_TAPS) && ((unsigned)q < (unsigned)L);
          const int v = live ? tok[live ? q : 0] : 8;
          const unsigned one = 0x3F80u << ((v & 1) << 4);   // bf16 1.0 in the low or the high half of a dword
          const int j = v >> 1;
          u32x4 oh;
          oh[0] = (j == 0) ? one : 0u;
          oh[1] = (j == 1) ? one : 0u;
          oh[2] = (j == 2) ? one : 0u;
          oh[3] = 0u;
          const bf16v8 bv = __builtin_bit_cast(bf16v8, oh);
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sw[ks][0], bv, s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sw[ks][1], bv, s1, 0, 0, 0);
        }
        acc[t][0] = s0;
        acc[t][1] = s1;
      }
    }
    ACCUM(7, t0);     // time-bias copy + stem MFMAs
#ifdef DRK_STAMPS
    epilogue_ln<NT, true, FAST>(smem, feat, acc, vecs, tbl, true, L, lane, ct, grp, tacc, t0);
#else
    epilogue_ln<NT, true, FAST>(smem, feat, acc, vecs, tbl, true, L, lane, ct, grp);
#endif
    if (a.dbg_feat && f == a.dump_fwd && a.dbg_layer == -1) dump_feat<NT>(a.dbg_feat, b, L, feat, lane, ct, grp);
    ACCUM(3, t0);

    // ---- n_layers x (dilated conv + ReLU + residual, then the next layer's LayerNorm) --------
    int m_next = reach_of(0);
    for (int layer = 0; layer < a.n_layers; ++layer) {
      const int dil = dil_of(layer);
      const int m = m_next;                                  // one integer division per layer, not per tap
      m_next = (layer + 1 < a.n_layers) ? reach_of(layer + 1) : 0;
      const int after = (layer + 1 < a.n_layers) ? 9 * (layer + 1) + 4 - m_next : head0_tap;   // the tap after this layer's last
      {   // the accumulators start at the conv bias (this lane's 8 channels): one add per value less in the epilogue
        f32x4 cbl[2];
        load_vec8(vecs + 384 * (layer + 1), 32 * ct + 4 * (opaque(lane) >> 4), cbl);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t][0] = cbl[0];
          acc[t][1] = cbl[1];
        }
      }
      const char* wn = (FAST ? wuni : wlane) + (long)(9 * layer + 4 - m + 1) * TAP_BYTES;   // FAST: uniform base + loff
      const char* wafter = (FAST ? wuni : wlane) + (long)after * TAP_BYTES;
      if constexpr (FAST) {
        if (dil == 64) {          // m == 3
          static_layer<NT, GRP, 64, (GRP == 1 ? 2 : 3)>(smem, wn, wafter, loff, ring, lane, acc);
        } else if (dil == 16) {   // m == 4
          static_layer<NT, GRP, 16, 4>(smem, wn, wafter, loff, ring, lane, acc);
        } else {                  // dilation <= 4: every shift stays within the zero padding
          const char* hcur = pad_base(smem, lane) - 4 * dil * ROWB;
          const int hstep = dil * ROWB;
          bf16v8 cur[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(hcur + RI(GRP, t, 0));
#ifndef DRK_NO_PRIO_FLIP
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
#else
          for (int k = 0; k < 8; ++k) {
            mma_tap_pad<NT, GRP, true>(hcur, hcur + hstep, wn, loff, ring, cur, acc);
            hcur += hstep;
            wn += TAP_BYTES;
          }
#endif
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
      ACCUM(0, t0);
#ifdef DRK_STAMPS
      epilogue_ln<NT, false, FAST>(smem, feat, acc, vecs + 384 * (layer + 1), tbl + 128 * (layer + 1), layer + 1 < a.n_layers,
                                   L, lane, ct, grp, tacc, t0);
#else
      epilogue_ln<NT, false, FAST>(smem, feat, acc, vecs + 384 * (layer + 1), tbl + 128 * (layer + 1), layer + 1 < a.n_layers,
                                   L, lane, ct, grp);
#endif
      if (a.dbg_feat && f == a.dump_fwd && a.dbg_layer == layer) dump_feat<NT>(a.dbg_feat, b, L, feat, lane, ct, grp);
      ACCUM(2, t0);
    }

    // ---- head: final_conv.0 (1x1) + ReLU, final_conv.2 (128 -> 5) ---------------------------
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
    // the taps after final_conv.0 are the first conv taps of the NEXT forward: their fragments arrive under the
    // head, the reverse step and the next stem
    if constexpr (FAST) {
      const char* hcur = pad_base(smem, lane);
      bf16v8 cur[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(hcur + RI(GRP, t, 0));
      mma_tap_pad<NT, GRP, false>(hcur, hcur, wuni + (long)first_tap * TAP_BYTES, loff, ring, cur, acc);
    } else {
      const char* bp[NT];
      bf16v8 cur[NT];
      tap_begin<NT>(smem, 0, L, lane, grp, bp, cur);
      mma_tap<NT, false>(smem, wlane + (long)first_tap * TAP_BYTES, ring, 0, L, lane, grp, bp, cur, acc);
    }
    __syncthreads();   // all reads of the feat image are done
    {
      const int lo2 = opaque(lane);
      const int pn2 = (lo2 & 15) ^ ((lo2 & 8) >> 1);
      const int ch2 = 32 * ct + 4 * (lo2 >> 4);
      f32x4 hb0[2];
      load_vec8(vecs + 384 * a.n_layers + 128, ch2, hb0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p = 16 * (grp + 2 * t) + pn2;
        f32x4 hv[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i) hv[s][i] = fmaxf(acc[t][s][i] + hb0[s][i], 0.f);
        if ((FAST && grp + 2 * t < 12) || p < L) store_h8(smem, p, ch2, hv);
      }
    }
    __syncthreads();
    {
      // 5 output rows (LDS-resident weights, rows 5..7 zero, columns >= 7 re-read row 7); position tiles dealt
      // round-robin to waves.  Same k-slot <-> channel map on both operands (goff).
      const int lo3 = opaque(lane);
      const int n = lo3 & 15, g = lo3 >> 4;
      const int pn = n ^ ((n & 8) >> 1);
      const int goff = 128 * (g & 1) + 64 * (g >> 1);
      for (int pt = wave; pt < ntiles; pt += NTHR / 64) {
        f32x4 lg = {0.f, 0.f, 0.f, 0.f};
        const int ar = (n < 7) ? n : 7;
        const char* abase = smem + OFF_HEAD2 + ar * HEAD2_ROWB + goff;
        const int p = 16 * pt + pn;
        const int row = (p < L) ? p : ZROW;
        const char* bbase = smem + OFF_HB + row * ROWB + goff;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16v8 av = *(const bf16v8*)(abase + 16 * ks);
          const bf16v8 bv = *(const bf16v8*)(bbase + 16 * ks);
          lg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, lg, 0, 0, 0);
        }
        const float* hb2 = vecs + 384 * a.n_layers + 256;
        if (g == 0) {
          logit[p * 8 + 0] = lg[0] + hb2[0];
          logit[p * 8 + 1] = lg[1] + hb2[1];
          logit[p * 8 + 2] = lg[2] + hb2[2];
          logit[p * 8 + 3] = lg[3] + hb2[3];
        } else if (g == 1) {
          logit[p * 8 + 4] = lg[0] + hb2[4];
        }
      }
    }
    __syncthreads();

    ACCUM(5, t0);     // head: final_conv.0 + ReLU + 128 -> 5, four barriers
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
    ACCUM(6, t0);     // reverse step
  }
  __syncthreads();
#ifdef DRK_STAMPS
  if (a.dbg_feat && lane == 0) {
    unsigned long long* o = (unsigned long long*)a.dbg_feat + (b * 8 + wave) * 4;
    o[0] = tacc[0]; o[1] = __builtin_amdgcn_s_memtime() - clk0; o[2] = tacc[2]; o[3] = tacc[3];
    ((unsigned long long*)a.dbg_feat)[(long)a.B * 33 + b * 8 + wave] = tacc[1];   // wait at the epilogue's first barrier
    ((unsigned long long*)a.dbg_feat)[(long)a.B * 41 + b * 8 + wave] = tacc[4];   // epilogue pass 1
    ((unsigned long long*)a.dbg_feat)[(long)a.B * 49 + b * 8 + wave] = tacc[5];   // head
    ((unsigned long long*)a.dbg_feat)[(long)a.B * 57 + b * 8 + wave] = tacc[6];   // reverse step
    ((unsigned long long*)a.dbg_feat)[(long)a.B * 65 + b * 8 + wave] = tacc[7];   // time-bias copy + stem MFMAs
    tacc[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    if (wave == 0) ((unsigned long long*)a.dbg_feat)[(long)a.B * 32 + b] = tacc[1];   // 100 MHz ticks of this workgroup
  }
#endif
  if (tid < L) a.x_out[b * L + tid] = tok[tid];
}

// 8 waves, two per SIMD (256 registers each): wave w = channel tile (w & 3) x position tiles (w >> 2) + 2t.
template <bool FAST>
__global__ void __launch_bounds__(512, 2) k_sample_bf16(const SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches, SGPR tile ids
  if (wave < 4)
    sample_body<7, 0, 512, FAST>(a, smem, tid, wave, lane);
  else if (FAST || a.L <= 208)
    sample_body<6, 1, 512, FAST>(a, smem, tid, wave, lane);
  else
    sample_body<7, 1, 512, false>(a, smem, tid, wave, lane);
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" int drk_sample_bf16(const void* wstream, const void* stem_w, const void* head2_w, const float* vecs, const float* tbias, long tb_fwd_stride,
                               const float* mc_table, const float* uniforms, const int64_t* x_init, int64_t* x_out,
                               signed char* traj_out, float* logits_out, int dump_fwd, float* dbg_feat, int dbg_layer, int B, int L, int num_steps, int n_layers,
                               const int* dilations, int noise_removal, uint64_t seed, uint64_t sample_offset,
                               int flags, void* stream) {
  if (B < 0 || L <= 0) return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: bad shape");
  if (flags & ~DRK_SAMPLE_GENERIC) return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: unknown flag");
  if (B == 0) return DRK_OK;
  if (L > MAXPOS) return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: L > 224 is not supported by the fused loop");
  if (n_layers < 1 || n_layers > MAX_LAYERS || num_steps < 0 || (num_steps == 0 && !noise_removal))
    return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: bad n_layers / num_steps");
  if (!wstream || !stem_w || !head2_w || !vecs || !tbias || !x_out || !dilations || (num_steps > 0 && !mc_table))
    return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: null pointer");
  SampleArgs a;
  a.wfrag = (const char*)wstream; a.stem_w = (const char*)stem_w; a.head2_w = (const char*)head2_w; a.vecs = vecs; a.tbias = tbias; a.tb_fwd_stride = tb_fwd_stride;
  a.mc_table = mc_table; a.uniforms = uniforms; a.x_init = x_init; a.x_out = x_out; a.traj_out = traj_out; a.logits_out = logits_out;
  a.dump_fwd = dump_fwd; a.dbg_feat = dbg_feat; a.dbg_layer = dbg_layer; a.B = B; a.L = L; a.num_steps = num_steps; a.n_layers = n_layers;
  a.noise_removal = noise_removal ? 1 : 0; a.seed = seed; a.sample_offset = sample_offset;
  for (int i = 0; i < 4; ++i) a.dilpk[i] = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (dilations[i] < 0 || dilations[i] > 255) return drk_set_error_msg(DRK_EINVAL, "drk_sample_bf16: dilation out of range");
    a.dilpk[i >> 3] |= (unsigned long long)dilations[i] << (8 * (i & 7));
  }
  bool fast = (L > 192 && L <= 208);   // the all-immediate-addressing build (see sample_body)
  for (int i = 0; i < n_layers; ++i) fast = fast && ((dilations[i] >= 1 && dilations[i] <= 4) || dilations[i] == 16 || dilations[i] == 64);
  if (flags & DRK_SAMPLE_GENERIC) fast = false;   // tests: the generic build must give the same bits
  {  // once per (device, build): lets the launch ask for more than 64 KB of LDS.  Not an allocation.
    static std::atomic<unsigned> lds_set[2][DRK_MAX_DEVICES];
    int devid = 0;
    hipError_t e = hipGetDevice(&devid);
    if (e != hipSuccess) return drk_set_error(e, "drk_sample_bf16(hipGetDevice)");
    if (devid < 0 || devid >= DRK_MAX_DEVICES || !lds_set[fast][devid].load(std::memory_order_acquire)) {
      const void* kern = fast ? (const void*)k_sample_bf16<true> : (const void*)k_sample_bf16<false>;
      e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) return drk_set_error(e, "drk_sample_bf16(hipFuncSetAttribute)");
      if (devid >= 0 && devid < DRK_MAX_DEVICES) lds_set[fast][devid].store(1u, std::memory_order_release);
    }
  }
  if (fast)
    hipLaunchKernelGGL(k_sample_bf16<true>, dim3(B), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_sample_bf16<false>, dim3(B), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
