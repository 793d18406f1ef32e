// Device helpers shared by the fused one-sequence-per-workgroup kernels (drk_sample_bf16.hip: the sampling loop;
// drk_train_fused_bf16.hip: the differentiable forward / backward of the fine-tune path): the LDS map, the MFMA tap
// loops (generic, zero-padded immediate addressing, static live ranges) and the small epilogue helpers.  See the
// header of drk_sample_bf16.hip for the work split and the lane <-> data maps.
#pragma once
#include "drk_common.h"
#include "../../include/drakes_hip.h"

#include <atomic>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define ROWB 272                 // bytes per h-image row: 128 bf16 + 16 B pad (bank slot = (row + chunk) mod 16); a
                                 // k-step is an IMMEDIATE offset (+16 B) from a per-lane row address: no address VALU
// The image geometry below is the sampler's and the fused train kernels' (one 200-bp sequence per workgroup);
// drk_group_bf16.hip (256-row images of long sequences) sets ZROW and DRK_LIVE_END before including this file and
// keeps its own LDS map behind the image.
#define MAXPOS 224               // 14 position tiles of 16
#ifndef ZROW
#define ZROW 224                 // the all-zero row of the h image (also the image's last row)
#endif
#ifndef DRK_LIVE_END
#define DRK_LIVE_END 208         // static live ranges: rows at and beyond this one are padding
#endif
#define TAP_BYTES 32768          // one tap of one layer in the fragment stream: [4 k-steps][4 ct][2 s][64 lanes][16 B]
#define KSTEP_BYTES 8192
#define SUB_BYTES 1024
#define STEM_BYTES 24576         // resident stem weights as MFMA A fragments: [3 k-steps][4 ct][2 s][64 lanes][16 B]
#define HEAD2_ROWB 272           // final_conv.2: [8 rows][128] bf16, rows 5..7 zero
#define MAX_LAYERS 20
#define HPAD 16                  // all-zero rows in front of the h image (logical rows -16..-1); rows L..223 stay zero too
#define OFF_HB (HPAD * ROWB)     // logical row 0
#define OFF_STEM (OFF_HB + (ZROW + 1) * ROWB)       // 65552
#define OFF_HEAD2 (OFF_STEM + STEM_BYTES)
#define PART_STRIDE 12                              // floats per position in the LayerNorm partial buffer (8 used): 48-byte
                                                    // rows make the two ds_read_b128 of a position conflict-free
#define OFF_PART (OFF_HEAD2 + 8 * HEAD2_ROWB)       // float [224][12]: slots 0-3 = per channel-tile sum, 4-7 = sum of squares
#define OFF_LOG (OFF_PART + MAXPOS * PART_STRIDE * 4)   // float [224][8]
#define OFF_TOK (OFF_LOG + MAXPOS * 8 * 4)          // int [224]
#define OFF_VECS (OFF_TOK + MAXPOS * 4)             // float [n_layers+1][3][128]
#define OFF_TB (OFF_VECS + (MAX_LAYERS + 1) * 3 * 128 * 4)   // float [n_layers][128]
#define LDS_BYTES (OFF_TB + MAX_LAYERS * 128 * 4)   // 153 744 of 163 840

// Returns `v` through an empty asm: everything derived from the result is recomputed where it is used instead of
// being hoisted out of the layer/step loops and kept (then spilled) for the whole kernel.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// The two waves of a SIMD take turns at issue priority, tap by tap, instead of "oldest first" for the whole MFMA phase
// (with oldest-first, waves 0-3 run ahead and waves 4-7 finish the phase alone, one wave per SIMD, then start their
// epilogue late): -4.6 % cycles per workgroup, +2.4 % wall in same-box A/B (round 2; -DDRK_NO_PRIO_FLIP builds without).
#ifndef DRK_NO_PRIO_FLIP
#define TAP_PRIO(hi) do { if (hi) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } while (0)
#else
#define TAP_PRIO(hi) do {} while (0)
#endif

// ---- one tap (or a 1x1 conv) ----------------------------------------------------------------
// Per-lane LDS addresses of the activation fragments (k-step 0) of the wave's NT tiles for a tap shifted by `off`.
template <int NT>
__device__ __forceinline__ void tap_rows(const char* smem, int off, int L, int lane, int grp, const char* (&bp)[NT]) {
  const int lo = opaque(lane);
  const int n = lo & 15, g = lo >> 4;
  const int pn = n ^ ((n & 8) >> 1);
  const int goff = 128 * (g & 1) + 64 * (g >> 1);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int pp = 16 * (grp + 2 * t) + pn + off;
    const int row = ((unsigned)pp < (unsigned)L) ? pp : ZROW;   // 'same' zero padding = the all-zero row
    bp[t] = smem + OFF_HB + row * ROWB + goff;
  }
}

// First tap of a layer: its k-step-0 fragments (the only LDS latency a layer exposes).
template <int NT>
__device__ __forceinline__ void tap_begin(const char* smem, int off, int L, int lane, int grp, const char* (&bp)[NT],
                                          bf16v8 (&cur)[NT]) {
  tap_rows<NT>(smem, off, L, lane, grp, bp);
#pragma unroll
  for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(bp[t]);
}

// acc[t][s] += W_tap[16-channel row tile s of this wave] . h[rows of tile t shifted by the tap], K = 128 = 4 k-steps.
// On entry cur[] holds the tap's k-step-0 activation fragments and bp[] their addresses.  ring[2ks+s] holds this tap's
// weight fragments (loaded one tap ago); as each k-step is consumed the SAME ring slots are refilled from `wnext`
// (this lane's pointer into the next live tap), so 8 global loads (one whole tap, ~2000 cycles) are always in flight.
// Activation fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks run; during the LAST k-step
// the reads are those of the NEXT tap's k-step 0 (HAS_NEXT, shift off_next), so taps follow each other without a
// bubble.
template <int NT, bool HAS_NEXT>
__device__ __forceinline__ void mma_tap(const char* smem, const char* __restrict__ wnext, bf16v8 (&ring)[8], int off_next,
                                        int L, int lane, int grp, const char* (&bp)[NT], bf16v8 (&cur)[NT],
                                        f32x4 (&acc)[NT][2]) {
  const char* bpn[NT];
  if (HAS_NEXT) tap_rows<NT>(smem, off_next, L, lane, grp, bpn);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16v8 a0 = ring[2 * ks], a1 = ring[2 * ks + 1];
    // Pinned order per k-step: two MFMAs (one activation fragment against both row tiles), then the LDS read that
    // REPLACES that fragment with the next k-step's (issued after its two readers, so no second buffer), two MFMAs,
    // a read, ...: the reads' issue is spread over the MFMA slots and each has a full k-step to land.  The refill of
    // the two ring slots comes after their last reader (no register copy).
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, cur[t], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, cur[t], acc[t][1], 0, 0, 0);
      if (ks + 1 < 4)
        cur[t] = *(const bf16v8*)(bp[t] + 16 * (ks + 1));
      else if (HAS_NEXT)
        cur[t] = *(const bf16v8*)(bpn[t]);
    }
    ring[2 * ks] = *(const bf16v8*)(wnext + ks * KSTEP_BYTES);
    ring[2 * ks + 1] = *(const bf16v8*)(wnext + ks * KSTEP_BYTES + SUB_BYTES);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (ks + 1 < 4 || HAS_NEXT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
  }
  if (HAS_NEXT) {
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = bpn[t];
  }
}

// ---- immediate addressing --------------------------------------------------------------------------------------
// The h image is physically zero-padded: HPAD all-zero rows in front of it, and rows L..223 are never written.  A shifted
// read that leaves [0, L) by at most HPAD rows then lands on zeros by itself, and the address of (tile t, shift off,
// k-step ks) is ONE per-lane base register plus the compile-time constant RI(grp, t, off) + 16 ks, which fits the 16-bit
// offset field of ds_read_b128: no per-tile address arithmetic, compare or select in the tap loops (they are bound by
// the SIMD's vector issue: every VALU instruction beside the MFMAs costs MFMA slots, MI355X_MICROARCH.md "ISSUE cost").
#define RI(grp, t, off) ((HPAD + 16 * ((grp) + 2 * (t)) + (off)) * ROWB)

// This lane's base: physical row 0 (logical row -HPAD) at its position-in-tile and k-slot group.
__device__ __forceinline__ const char* pad_base(const char* smem, int lane) {
  const int lo = opaque(lane);
  const int pn = (lo & 15) ^ ((lo & 8) >> 1);
  return smem + (OFF_HB - HPAD * ROWB) + pn * ROWB + 128 * ((lo >> 4) & 1) + 64 * (lo >> 5);
}

// Weight fragment load with a wave-uniform base (SGPR pair) and a 32-bit per-lane offset: the compiler then uses the
// scalar-base form of global_load and moves tap / k-step strides to the scalar unit, instead of 64-bit VALU adds.
__device__ __forceinline__ bf16v8 wload(const char* __restrict__ wu, unsigned loff) {
  return *(const bf16v8*)(wu + (size_t)loff);
}

// mma_tap for dilation <= 4 (every shift within HPAD rows): hcur / hnext = pad_base shifted by this / the next tap.
template <int NT, int GRP, bool HAS_NEXT>
__device__ __forceinline__ void mma_tap_pad(const char* hcur, const char* hnext, const char* __restrict__ wnext, unsigned loff,
                                            bf16v8 (&ring)[8], bf16v8 (&cur)[NT], f32x4 (&acc)[NT][2]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16v8 a0 = ring[2 * ks], a1 = ring[2 * ks + 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, cur[t], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, cur[t], acc[t][1], 0, 0, 0);
      if (ks + 1 < 4)
        cur[t] = *(const bf16v8*)(hcur + RI(GRP, t, 0) + 16 * (ks + 1));
      else if (HAS_NEXT)
        cur[t] = *(const bf16v8*)(hnext + RI(GRP, t, 0));
    }
    ring[2 * ks] = wload(wnext + ks * KSTEP_BYTES, loff);
    ring[2 * ks + 1] = wload(wnext + ks * KSTEP_BYTES + SUB_BYTES, loff);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (ks + 1 < 4 || HAS_NEXT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
  }
}

// ---- layers with dead tiles: straight-line taps with COMPILE-TIME live ranges ------------------------------------
// With dilation 16 and 64 many (tile, tap) pairs multiply nothing but padding: a tile whose 16 shifted rows all fall
// outside [0, L) (53 % of the pairs at dilation 64, 17 % at 16, 12 % of a whole forward).  Skipping them under a
// runtime test does not pay (a wave-uniform branch per tile makes the compiler drain the LDS queue at every join;
// a switch over variants spills).  But for 192 < L <= 208 and shifts that are multiples of 16 the live tiles of every
// tap are known at compile time: tile T (positions 16T..16T+15) is live for shift `off` iff 16T + off < 208 (off > 0)
// or 16T + 15 >= -off (off < 0).  The dead tiles of a wave are a prefix or a suffix of its tiles, so a tap is a static
// range [T0, T1), and a layer is 2m+1 different instantiations in a row: no branch, no join, exact LDS-read counts.
__host__ __device__ constexpr int live_lo(int off, int grp, int nt) {   // first live t of the wave's tiles T = grp + 2t
  int t = 0;
  while (t < nt && off < 0 && 16 * (grp + 2 * t) + 15 < -off) ++t;
  return t;
}
__host__ __device__ constexpr int live_hi(int off, int grp, int nt) {   // one past the last live t
  int t = nt;
  while (t > 0 && off > 0 && 16 * (grp + 2 * (t - 1)) + off >= DRK_LIVE_END) --t;
  return t < live_lo(off, grp, nt) ? live_lo(off, grp, nt) : t;
}

// One tap (shift OFF) over the static tile range [T0, T1); during the last k-step the k-step-0 fragments of the NEXT
// tap's range [N0, N1) are fetched (shift OFFN).  hb = pad_base: a live tile's shifted rows are rows of the image (or
// its zero rows 200..207), so every address is hb + a constant.  Otherwise as mma_tap.
template <int NT, int GRP, int OFF, int T0, int T1, int OFFN, int N0, int N1>
__device__ __forceinline__ void mma_tap_static(const char* hb, const char* __restrict__ wnext, unsigned loff, bf16v8 (&ring)[8],
                                               bf16v8 (&cur)[NT], f32x4 (&acc)[NT][2]) {
  if constexpr (T1 - T0 >= 1 && T1 - T0 <= 2) {
    // One or two live tiles: 2-4 MFMAs per k-step cannot cover an LDS read, so all of the tap's fragments are
    // requested up front (the k-step-0 ones are in cur[] already) and one latency is paid instead of three.
    constexpr int NL = T1 - T0;
    bf16v8 f[3][NL];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int t = 0; t < NL; ++t) f[k][t] = *(const bf16v8*)(hb + RI(GRP, T0 + t, OFF) + 16 * (k + 1));
#pragma unroll
    for (int t = N0; t < N1; ++t)
      if (t < T0 || t >= T1) cur[t] = *(const bf16v8*)(hb + RI(GRP, t, OFFN));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16v8 a0 = ring[2 * ks], a1 = ring[2 * ks + 1];
#pragma unroll
      for (int t = 0; t < NL; ++t) {
        const bf16v8 b = (ks == 0) ? cur[T0 + t] : f[ks - 1][t];
        acc[T0 + t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc[T0 + t][0], 0, 0, 0);
        acc[T0 + t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc[T0 + t][1], 0, 0, 0);
      }
      ring[2 * ks] = wload(wnext + ks * KSTEP_BYTES, loff);
      ring[2 * ks + 1] = wload(wnext + ks * KSTEP_BYTES + SUB_BYTES, loff);
    }
#pragma unroll
    for (int t = T0; t < T1; ++t)
      if (t >= N0 && t < N1) cur[t] = *(const bf16v8*)(hb + RI(GRP, t, OFFN));
    return;
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16v8 a0 = ring[2 * ks], a1 = ring[2 * ks + 1];
    if (ks + 1 == 4) {   // tiles that only the next tap uses
#pragma unroll
      for (int t = N0; t < N1; ++t)
        if (t < T0 || t >= T1) cur[t] = *(const bf16v8*)(hb + RI(GRP, t, OFFN));
    }
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, cur[t], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, cur[t], acc[t][1], 0, 0, 0);
      if (ks + 1 < 4) {
        cur[t] = *(const bf16v8*)(hb + RI(GRP, t, OFF) + 16 * (ks + 1));
      } else if (t >= N0 && t < N1) {
        cur[t] = *(const bf16v8*)(hb + RI(GRP, t, OFFN));
      }
    }
    ring[2 * ks] = wload(wnext + ks * KSTEP_BYTES, loff);
    ring[2 * ks + 1] = wload(wnext + ks * KSTEP_BYTES + SUB_BYTES, loff);
    // pinned issue order, as in mma_tap: (reads of tiles only the next tap uses,) then per live tile two MFMAs and the
    // read that replaces its fragment, then the two ring refills
    if (ks + 1 == 4) {
      constexpr int extra = (N1 - N0) - ((T1 < N1 ? T1 : N1) - (T0 > N0 ? T0 : N0) > 0 ? (T1 < N1 ? T1 : N1) - (T0 > N0 ? T0 : N0) : 0);
      if (extra > 0) __builtin_amdgcn_sched_group_barrier(0x100, extra, 0);
    }
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      if (ks + 1 < 4 || (t >= N0 && t < N1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
  }
}

// Taps K .. M of a layer with static dilation D (shift K*D), each with its own live range; wn = this lane's pointer to
// the weights of tap K+1, wafter = to the tap that follows the layer's last.
template <int NT, int GRP, int D, int M, int K>
__device__ __forceinline__ void static_taps(const char* hb, const char* wn, const char* wafter, unsigned loff, bf16v8 (&ring)[8],
                                            bf16v8 (&cur)[NT], f32x4 (&acc)[NT][2]) {
  constexpr int off = K * D, offn = (K + 1) * D;
  constexpr bool last = (K == M);
  constexpr int T0 = live_lo(off, GRP, NT), T1 = live_hi(off, GRP, NT);
  constexpr int N0 = last ? 0 : live_lo(offn, GRP, NT), N1 = last ? 0 : live_hi(offn, GRP, NT);
  TAP_PRIO(((K + M + GRP) & 1) != 0);
  mma_tap_static<NT, GRP, off, T0, T1, offn, N0, N1>(hb, last ? wafter : wn, loff, ring, cur, acc);
  if constexpr (!last) static_taps<NT, GRP, D, M, K + 1>(hb, wn + TAP_BYTES, wafter, loff, ring, cur, acc);
}

// Whole MMA phase of such a layer: first tap's fragments (the only exposed LDS latency), then the taps.
template <int NT, int GRP, int D, int M>
__device__ __forceinline__ void static_layer(const char* smem, const char* wn, const char* wafter, unsigned loff, bf16v8 (&ring)[8],
                                             int lane, f32x4 (&acc)[NT][2]) {
  constexpr int T0 = live_lo(-M * D, GRP, NT), T1 = live_hi(-M * D, GRP, NT);
  // taps beyond +-M must have no live tile for this wave (the caller skips them, weights included)
  static_assert(M == 4 || (live_lo(-(M + 1) * D, GRP, NT) >= live_hi(-(M + 1) * D, GRP, NT) &&
                           live_lo((M + 1) * D, GRP, NT) >= live_hi((M + 1) * D, GRP, NT)) || (M + 1) * D >= DRK_LIVE_END,
                "skipped taps must be dead");
  const char* hb = pad_base(smem, lane);
  bf16v8 cur[NT];
#pragma unroll
  for (int t = T0; t < T1; ++t) cur[t] = *(const bf16v8*)(hb + RI(GRP, t, -M * D));
  if (T1 > T0) __builtin_amdgcn_sched_group_barrier(0x100, T1 - T0, 0);   // the k-step-0 reads first, as a group
  static_taps<NT, GRP, D, M, -M>(hb, wn, wafter, loff, ring, cur, acc);
}

// This lane's channels: 32 ct + 16 s + 4 g + i (s = 0,1; i = 0..3) <-> register [s][i].
__device__ __forceinline__ void load_vec8(const float* v, int ch0, f32x4 (&o)[2]) {
  o[0] = *(const f32x4*)(v + ch0);
  o[1] = *(const f32x4*)(v + ch0 + 16);
}

// Writes this lane's 8 channel values of position p as bf16 into the h image (natural channel order).
__device__ __forceinline__ void store_h8(char* smem, int p, int ch0, const f32x4 (&x)[2]) {
  char* row = smem + OFF_HB + p * ROWB + 2 * ch0;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16v4 pk;
    pk[0] = (__bf16)x[s][0]; pk[1] = (__bf16)x[s][1]; pk[2] = (__bf16)x[s][2]; pk[3] = (__bf16)x[s][3];
    *(bf16v4*)(row + 32 * s) = pk;
  }
}
