// drk_conv_fwd_x3 — the residual layer's convolution (models/dnaconv.py:96-98) at fp32 GRADE on the bf16 matrix cores:
// every fp32 operand is written as the sum of three bf16 numbers, x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1): 3 x 8 significant bits cover fp32's 24), and a product w * h is taken as the six partial
// products whose planes add up to at most 2 (w0h0, w0h1, w1h0, w0h2, w1h1, w2h0); the dropped ones are <= 2^-24 of the
// product, one fp32 rounding.  bf16 x bf16 is exact in the fp32 accumulator, so the result differs from an fp32
// convolution by summation order and those dropped terms only.  Six v_mfma_f32_16x16x32_bf16 (2.5 PFLOP/s dense) do the
// work of one pass of the f32-input MFMA (157 TFLOP/s): 0.375 x its time at equal pipe occupancy.
//
// Same interface as drk_conv_fwd_f32: fp32 h in, fp32 out, bias + ReLU (+ mask) + residual fused.  The weights arrive
// split (three bf16 fragment streams, fastpath._frag_tap order, packed once per weight update); the activations are
// split by the thread that stages them, ONCE per element and tap, on their way from global memory into LDS, which holds
// the three planes of a tap's 128 rows (splitting at the MFMA operand read instead — every element four times, once per
// channel-tile wave — left the matrix pipe at 34 %: 44 VALU operations per 12 MFMAs saturate the vector issue port):
// h stays fp32 in memory and the LayerNorm kernel stays the exact one.
//
// Workgroup = 512 threads = 8 waves (two per SIMD) on a tile of 128 rows x 128 output channels.  Wave (ct, kh) owns
// output channels [32 ct, 32 ct + 32), ALL 128 rows, and input-channel half kh (k-steps 2 kh, 2 kh + 1): no weight
// fragment is fetched twice inside a workgroup (the L1 does not merge duplicate requests, and three planes make the
// weight stream 864 KB per layer).  After the tap loop the two halves of a tile meet through LDS and each wave finishes
// 64 rows.  Each tap's 128 shifted rows are fetched as whole 512-byte rows, in flight in registers under the previous
// tap's 192 MFMAs per wave, then split and parked ([row][plane][128 bf16], 784-byte rows; one buffer — two do not fit —
// so two barriers per tap around the parking); rows whose shifted position leaves [0, L) are staged as zeros ('same'
// padding and the sequence boundary in one test).
// MFMA-bound: 6 x 295 kFLOP per row against 512 B x 9 taps of L2-resident re-reads + 864 KB of weights per 128 rows.
#include <atomic>
#include "drk_common.h"
#include "../../include/drakes_hip.h"

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));

#define X3_NT 6                          // 16-row tiles per workgroup tile
#define X3_ROWS (16 * X3_NT)             // 96: two buffers of 96 rows x three planes fit the LDS, two of 128 do not
#define X3_ROWB 784                      // 3 planes x 128 bf16 + 16 B pad
#define X3_BUF (X3_ROWS * X3_ROWB)       // 75 264
#define X3_PLANE_BYTES 32768             // one plane of one tap: [4 k-steps][4 ct][2 s][64 lanes][16 B]
#define X3_TAP_BYTES (3 * X3_PLANE_BYTES)
#define X3_KSTEP_BYTES 8192
#define X3_SUB_BYTES 1024

// -DX3_STAMPS (tools/x3_stamps.py, never shipped): thread 0 stamps the phases of each workgroup's SECOND tile into `mask`
#ifdef X3_STAMPS
#define X3_STAMP(i) do { if (threadIdx.x == 0 && tile == t_begin + 1) ((unsigned long long*)a.mask)[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define X3_STAMP(i) do { } while (0)
#endif

struct X3Args {
  const float* h;           // fp32 [M][128]
  const char* w3;           // [taps][3 planes][32 KB] bf16 fragments
  const float* bias;        // [128] or NULL
  const float* resid;       // fp32 [M][128] or NULL (may alias out)
  float* out;               // fp32 [M][128]
  unsigned char* mask;      // optional uint8 [M][128]
  long M;
  int L, taps, dil, first_tap, relu;
};

typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));

// x (4 floats) -> three bf16 planes
__device__ __forceinline__ void split3(const f32x4 x, bf16v4& p0, bf16v4& p1, bf16v4& p2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 a = (__bf16)x[j];
    const float r1 = x[j] - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    p0[j] = a;
    p1[j] = b;
    p2[j] = (__bf16)r2;
  }
}

__global__ void __launch_bounds__(512, 2) k_conv_x3(const X3Args a, const int tiles_per_wg, const int ntiles) {
  __shared__ __attribute__((aligned(16))) char smem[2 * X3_BUF];
  // one workgroup per CU (147 KB of LDS) walks a contiguous run of tiles; (tile, tap) pairs form ONE stream of staged row
  // sets: the rows of the next pair are fetched at the start of this pair's MFMAs and split + parked into the other
  // buffer half-way through them, so the vector work of the split runs beside the matrix pipe and there is one barrier
  // per tap
  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = t_begin + tiles_per_wg < ntiles ? t_begin + tiles_per_wg : ntiles;
  if (t_begin >= t_end) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ct = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int kh = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int n = lane & 15, g = lane >> 4;
  const int ch0 = 32 * ct + 4 * g;                       // this lane's 4 output channels of row tile s: ch0 + 16 s ..
  const int L = a.L;
  const int center = a.taps / 2;
  const int last_tap = a.taps - 1 - a.first_tap;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // staging role: rows srow + 16 i (i < X3_NT), 16-byte chunk scol of the 512-byte row
  const int srow = tid >> 5, scol = tid & 31;
  f32x4 stage[X3_NT];
  auto positions = [&](long r0, int (&sp)[X3_NT]) {
    const unsigned p0 = (unsigned)(r0 % L);
#pragma unroll
    for (int i = 0; i < X3_NT; ++i) sp[i] = (int)((p0 + srow + 16 * i) % (unsigned)L);
  };
  // Loads are unconditional on clamped addresses and the rows that must read as zero are blanked when they are parked:
  // a zero-fill + predicated load here makes the compiler wait for EVERY load in flight (the weight fragments requested
  // just before) ahead of the fill, once per tap.
  unsigned live = 0;
  auto fetch = [&](long r0, const int (&sp)[X3_NT], int off) {
    live = 0;
#pragma unroll
    for (int i = 0; i < X3_NT; ++i) {
      const long row = r0 + srow + 16 * i;
      const int p = sp[i] + off;
      const bool ok = row < a.M && p >= 0 && p < L;
      live |= ok ? (1u << i) : 0u;
      long src = row + off;
      src = src < 0 ? 0 : src >= a.M ? a.M - 1 : src;
      stage[i] = *(const f32x4*)(a.h + src * DRK_C + 4 * scol);
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int i = 0; i < X3_NT; ++i) {
      bf16v4 q0, q1, q2;
      split3((live >> i) & 1 ? stage[i] : zero4, q0, q1, q2);
      char* d = smem + buf * X3_BUF + (srow + 16 * i) * X3_ROWB + 8 * scol;
      *(bf16v4*)(d) = q0;
      *(bf16v4*)(d + 256) = q1;
      *(bf16v4*)(d + 512) = q2;
    }
  };

  // A fragments of one k-step: [plane][s]; the ring holds the k-step in use and the next one
  const char* wlane = a.w3 + (2 * kh) * X3_KSTEP_BYTES + ct * (2 * X3_SUB_BYTES) + lane * 16;
  bf16v8 wa[2][3][2];
  auto load_w = [&](int slot, int tap, int ksl) {
    const char* p = wlane + (long)tap * X3_TAP_BYTES + ksl * X3_KSTEP_BYTES;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      wa[slot][pl][0] = *(const bf16v8*)(p + pl * X3_PLANE_BYTES);
      wa[slot][pl][1] = *(const bf16v8*)(p + pl * X3_PLANE_BYTES + X3_SUB_BYTES);
    }
  };
  int spos[X3_NT];
  positions((long)t_begin * X3_ROWS, spos);
  load_w(0, a.first_tap, 0);
  fetch((long)t_begin * X3_ROWS, spos, (a.first_tap - center) * a.dil);
  park(0);
  __syncthreads();

  // B operand: lane (n, g), k-step ks holds input channels 64 (g & 1) + 32 (g >> 1) + 8 ks + 0..7 of position n
  const int boff = n * X3_ROWB + 2 * (64 * (g & 1) + 32 * (g >> 1) + 16 * kh);
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const long r0 = (long)tile * X3_ROWS;
    const bool next_tile = tile + 1 < t_end;
    f32x4 acc[X3_NT][2];
#pragma unroll
    for (int t = 0; t < X3_NT; ++t) {
      acc[t][0] = zero4;
      acc[t][1] = zero4;
    }
    X3_STAMP(0);
    for (int tap = a.first_tap; tap <= last_tap; ++tap) {
      const bool more = tap < last_tap;
      const bool staged = more || next_tile;
      // requested in this order: loads return in order, and the second k-step's fragments are needed long before the
      // next rows
      load_w(1, tap, 1);
      if (more) {
        fetch(r0, spos, (tap + 1 - center) * a.dil);
      } else if (next_tile) {
        positions(r0 + X3_ROWS, spos);
        fetch(r0 + X3_ROWS, spos, (a.first_tap - center) * a.dil);
      }
      const char* const base = smem + buf * X3_BUF + boff;
#pragma unroll
      for (int ksl = 0; ksl < 2; ++ksl) {
        if (ksl == 1) {
          load_w(0, more ? tap + 1 : a.first_tap, 0);    // the next tap's (or the next tile's) first k-step
          if (staged) park(buf ^ 1);                     // the other buffer: its readers finished before the last barrier
        }
        bf16v8 b0 = *(const bf16v8*)(base + 16 * ksl), b1 = *(const bf16v8*)(base + 16 * ksl + 256),
               b2 = *(const bf16v8*)(base + 16 * ksl + 512);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int t = 0; t < X3_NT; ++t) {
          const bf16v8 c0 = b0, c1 = b1, c2 = b2;
          if (t + 1 < X3_NT) {
            const char* nb = base + (16 * (t + 1)) * X3_ROWB + 16 * ksl;
            b0 = *(const bf16v8*)(nb);
            b1 = *(const bf16v8*)(nb + 256);
            b2 = *(const bf16v8*)(nb + 512);
          }
          // two accumulation chains side by side (a dependent v_mfma waits ~40 cycles for its predecessor, an
          // independent one issues after 16); small terms first
          f32x4 e0 = acc[t][0], e1 = acc[t][1];
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][2][0], c0, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][2][1], c0, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][1][0], c1, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][1][1], c1, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][0], c2, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][1], c2, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][1][0], c0, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][1][1], c0, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][0], c1, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][1], c1, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][0], c0, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][1], c0, e1, 0, 0, 0);
          acc[t][0] = e0;
          acc[t][1] = e1;
          // the next row tile's three fragments are read BEFORE this tile's 12 MFMAs issue (left alone, the compiler
          // sinks the reads to just before their first use and every tile waits out an LDS round trip)
          if (t + 1 < X3_NT) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
      }
      X3_STAMP(1 + 2 * (tap - a.first_tap > 3 ? 4 : tap - a.first_tap));
      __syncthreads();         // this tap's rows have been read by every wave, the next ones are parked
      X3_STAMP(2 + 2 * (tap - a.first_tap > 3 ? 4 : tap - a.first_tap));
      if (staged) buf ^= 1;
    }

    // ---- the two input-channel halves meet through the buffer that holds no staged rows: kh 0 finishes the first
    // X3_NT/2 row tiles, kh 1 the others -----------------------------------------------------------------------------------
    constexpr int HT = X3_NT / 2;
    // bias and residual of this wave's output rows: requested now, they arrive under the exchange (out may be resid, so
    // loads left inside the store loop below would each wait out a round trip)
    const f32x4 cb0 = a.bias ? *(const f32x4*)(a.bias + ch0) : zero4, cb1 = a.bias ? *(const f32x4*)(a.bias + ch0 + 16) : zero4;
    f32x4 res[HT][2];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      long row = r0 + 16 * HT * kh + 16 * t + n;
      row = row < a.M ? row : a.M - 1;
      res[t][0] = a.resid ? *(const f32x4*)(a.resid + row * DRK_C + ch0) : zero4;
      res[t][1] = a.resid ? *(const f32x4*)(a.resid + row * DRK_C + ch0 + 16) : zero4;
    }
    f32x4 mine[HT][2];
    {
      char* const xbuf = smem + (next_tile ? (buf ^ 1) : buf) * X3_BUF;
      f32x4* const xw = (f32x4*)xbuf + (tid >> 6) * (2 * HT * 64) + lane;     // [wave][2 HT][64 lanes]
      const f32x4* const xr = (const f32x4*)xbuf + ((tid >> 6) ^ 4) * (2 * HT * 64) + lane;
      if (kh == 0) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          xw[(2 * t) * 64] = acc[HT + t][0];
          xw[(2 * t + 1) * 64] = acc[HT + t][1];
        }
      } else {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          xw[(2 * t) * 64] = acc[t][0];
          xw[(2 * t + 1) * 64] = acc[t][1];
        }
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          mine[t][0] = acc[t][0] + xr[(2 * t) * 64];
          mine[t][1] = acc[t][1] + xr[(2 * t + 1) * 64];
        }
      } else {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          mine[t][0] = xr[(2 * t) * 64] + acc[HT + t][0];                     // (half 0) + (half 1) on both sides
          mine[t][1] = xr[(2 * t + 1) * 64] + acc[HT + t][1];
        }
      }
    }
    X3_STAMP(11);
    // ---- epilogue: bias, ReLU (+ mask), residual ---------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const long row = r0 + 16 * HT * kh + 16 * t + n;
      if (row >= a.M) continue;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 r = mine[t][s] + (s ? cb1 : cb0);
        if (a.relu) {
#ifndef X3_STAMPS
          if (a.mask) {
            const unsigned m = (r[0] > 0.f ? 1u : 0u) | (r[1] > 0.f ? 0x100u : 0u) | (r[2] > 0.f ? 0x10000u : 0u) |
                               (r[3] > 0.f ? 0x1000000u : 0u);
            *(unsigned*)(a.mask + row * DRK_C + ch0 + 16 * s) = m;
          }
#endif
#pragma unroll
          for (int i = 0; i < 4; ++i) r[i] = fmaxf(r[i], 0.f);
        }
        r += res[t][s];
        *(f32x4*)(a.out + row * DRK_C + ch0 + 16 * s) = r;
      }
    }
    X3_STAMP(12);
    if (next_tile) __syncthreads();   // the exchange buffer is free again before the next tile's second tap parks into it
  }
}

// ---- small dilations (96 + 8 * dilation <= 128 rows: dilation <= 4 with nine taps; 1x1 convs): the tile's rows and
// their halo are split and parked ONCE, a tap is a row offset into that image, and a lane whose shifted position leaves
// [0, L) reads an all-zero row instead.  No staging, split or barrier inside the tap loop; the next tile's rows are in
// flight in registers under the whole tap loop and are parked after the two halves have met.
// NT = 16-row tiles per workgroup tile: 6 (96 rows, as the per-tap kernel) or 8 (128 rows: for launches of few tiles —
// the fine-tune shapes — where 128-row tiles need one round of the chip and 96-row tiles two).  Image rows: 16 NT + 32.
#define X3H_HALO_MAX 16
// HM = the largest halo the image has room for: 16 rows (dilation <= 4) with 96- or 128-row tiles, or 64 rows (dilation
// 16) with 64-row tiles (192 image rows, 150.5 KB).
template <int NT, int HM = X3H_HALO_MAX>
__global__ void __launch_bounds__(512, 2) k_conv_x3_halo(const X3Args a, const int tiles_per_wg, const int ntiles) {
  constexpr int TROWS = 16 * NT;                         // rows per tile
  constexpr int X3H_ROWS = TROWS + 2 * HM;               // image rows
  constexpr int X3H_ZERO = X3H_ROWS * X3_ROWB;           // the all-zero row
  constexpr int X3H_SMEM = X3H_ZERO + X3_ROWB;           // 101 136 B (NT = 6), 126 224 B (NT = 8)
  constexpr int NST = X3H_ROWS / 16;                     // staging passes of 16 rows
  __shared__ __attribute__((aligned(16))) char smem[X3H_SMEM];
  const int t_begin = blockIdx.x * tiles_per_wg;
  const int t_end = t_begin + tiles_per_wg < ntiles ? t_begin + tiles_per_wg : ntiles;
  if (t_begin >= t_end) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int ct = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int kh = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int n = lane & 15, g = lane >> 4;
  const int ch0 = 32 * ct + 4 * g;
  const int L = a.L;
  const int center = a.taps / 2;
  const int last_tap = a.taps - 1 - a.first_tap;
  const int halo = center * a.dil;                       // <= HM, checked by the entry point
  const int rows_total = TROWS + 2 * halo;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // staging role: image rows srow + 16 i (i < NST), 16-byte chunk scol of the 512-byte row
  const int srow = tid >> 5, scol = tid & 31;
  f32x4 stage[NST];
  unsigned live = 0;
  auto fetch = [&](long r0) {
    live = 0;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const long row = r0 - halo + srow + 16 * i;
      const bool ok = row >= 0 && row < a.M;               // (rows of a neighbouring sequence are masked at the read)
      live |= ok ? (1u << i) : 0u;
      const long src = row < 0 ? 0 : row >= a.M ? a.M - 1 : row;
      stage[i] = *(const f32x4*)(a.h + src * DRK_C + 4 * scol);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      if (srow + 16 * i < rows_total) {
        bf16v4 q0, q1, q2;
        split3((live >> i) & 1 ? stage[i] : zero4, q0, q1, q2);
        char* d = smem + (srow + 16 * i) * X3_ROWB + 8 * scol;
        *(bf16v4*)(d) = q0;
        *(bf16v4*)(d + 256) = q1;
        *(bf16v4*)(d + 512) = q2;
      }
    }
  };
  if (tid < X3_ROWB / 16) *(f32x4*)(smem + X3H_ZERO + 16 * tid) = zero4;

  const char* wlane = a.w3 + (2 * kh) * X3_KSTEP_BYTES + ct * (2 * X3_SUB_BYTES) + lane * 16;
  bf16v8 wa[2][3][2];
  auto load_w = [&](int slot, int tap, int ksl) {
    const char* p = wlane + (long)tap * X3_TAP_BYTES + ksl * X3_KSTEP_BYTES;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      wa[slot][pl][0] = *(const bf16v8*)(p + pl * X3_PLANE_BYTES);
      wa[slot][pl][1] = *(const bf16v8*)(p + pl * X3_PLANE_BYTES + X3_SUB_BYTES);
    }
  };
  load_w(0, a.first_tap, 0);
  fetch((long)t_begin * TROWS);
  park();
  __syncthreads();

  const int koff = 2 * (64 * (g & 1) + 32 * (g >> 1) + 16 * kh);     // this lane's 16-byte chunk of a plane, k-step 2 kh
  for (int tile = t_begin; tile < t_end; ++tile) {
    const long r0 = (long)tile * TROWS;
    const bool next_tile = tile + 1 < t_end;
    const unsigned p0 = (unsigned)(r0 % L);
    int pos[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pos[t] = (int)((p0 + 16 * t + n) % (unsigned)L);
    f32x4 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t][0] = zero4;
      acc[t][1] = zero4;
    }
    load_w(1, a.first_tap, 1);
    if (next_tile) fetch(r0 + TROWS);                    // in flight under the whole tap loop
    for (int tap = a.first_tap; tap <= last_tap; ++tap) {
      const bool more = tap < last_tap;
      const int sh = (tap - center) * a.dil;
      int addr[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        addr[t] = (unsigned)(pos[t] + sh) < (unsigned)L ? (halo + 16 * t + n + sh) * X3_ROWB + koff : X3H_ZERO + koff;
#pragma unroll
      for (int ksl = 0; ksl < 2; ++ksl) {
        if (tap > a.first_tap || ksl == 1) {               // (the first k-step's successor was requested ahead of the rows)
          if (ksl == 0) load_w(1, tap, 1);
          else load_w(0, more ? tap + 1 : a.first_tap, 0);
        }
        bf16v8 b0 = *(const bf16v8*)(smem + addr[0] + 16 * ksl), b1 = *(const bf16v8*)(smem + addr[0] + 16 * ksl + 256),
               b2 = *(const bf16v8*)(smem + addr[0] + 16 * ksl + 512);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bf16v8 c0 = b0, c1 = b1, c2 = b2;
          if (t + 1 < NT) {
            const char* nb = smem + addr[t + 1] + 16 * ksl;
            b0 = *(const bf16v8*)(nb);
            b1 = *(const bf16v8*)(nb + 256);
            b2 = *(const bf16v8*)(nb + 512);
          }
          f32x4 e0 = acc[t][0], e1 = acc[t][1];
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][2][0], c0, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][2][1], c0, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][1][0], c1, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][1][1], c1, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][0], c2, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][1], c2, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][1][0], c0, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][1][1], c0, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][0], c1, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][1], c1, e1, 0, 0, 0);
          e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][0], c0, e0, 0, 0, 0);
          e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ksl][0][1], c0, e1, 0, 0, 0);
          acc[t][0] = e0;
          acc[t][1] = e1;
          if (t + 1 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
      }
    }
    // the residual is requested as soon as the last MFMA has issued: it arrives under the exchange and the parking
    f32x4 res[NT / 2][2];
#pragma unroll
    for (int t = 0; t < NT / 2; ++t) {
      long row = r0 + 16 * (NT / 2) * kh + 16 * t + n;
      row = row < a.M ? row : a.M - 1;
      res[t][0] = a.resid ? *(const f32x4*)(a.resid + row * DRK_C + ch0) : zero4;
      res[t][1] = a.resid ? *(const f32x4*)(a.resid + row * DRK_C + ch0 + 16) : zero4;
    }
    __syncthreads();           // every wave is past its reads of the image: it carries the exchange now
    constexpr int HT = NT / 2;
    f32x4 mine[HT][2];
    {
      f32x4* const xw = (f32x4*)smem + (tid >> 6) * (2 * HT * 64) + lane;
      const f32x4* const xr = (const f32x4*)smem + ((tid >> 6) ^ 4) * (2 * HT * 64) + lane;
      if (kh == 0) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          xw[(2 * t) * 64] = acc[HT + t][0];
          xw[(2 * t + 1) * 64] = acc[HT + t][1];
        }
      } else {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          xw[(2 * t) * 64] = acc[t][0];
          xw[(2 * t + 1) * 64] = acc[t][1];
        }
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          mine[t][0] = acc[t][0] + xr[(2 * t) * 64];
          mine[t][1] = acc[t][1] + xr[(2 * t + 1) * 64];
        }
      } else {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          mine[t][0] = xr[(2 * t) * 64] + acc[HT + t][0];
          mine[t][1] = xr[(2 * t + 1) * 64] + acc[HT + t][1];
        }
      }
    }
    if (next_tile) {
      __syncthreads();         // the exchange has been read: the image takes the next tile's rows
      park();
    }
    const f32x4 cb0 = a.bias ? *(const f32x4*)(a.bias + ch0) : zero4, cb1 = a.bias ? *(const f32x4*)(a.bias + ch0 + 16) : zero4;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const long row = r0 + 16 * HT * kh + 16 * t + n;
      if (row >= a.M) continue;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 r = mine[t][s] + (s ? cb1 : cb0);
        if (a.relu) {
          if (a.mask) {
            const unsigned m = (r[0] > 0.f ? 1u : 0u) | (r[1] > 0.f ? 0x100u : 0u) | (r[2] > 0.f ? 0x10000u : 0u) |
                               (r[3] > 0.f ? 0x1000000u : 0u);
            *(unsigned*)(a.mask + row * DRK_C + ch0 + 16 * s) = m;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) r[i] = fmaxf(r[i], 0.f);
        }
        r += res[t][s];
        *(f32x4*)(a.out + row * DRK_C + ch0 + 16 * s) = r;
      }
    }
    if (next_tile) __syncthreads();   // the parked rows are visible
  }
}

extern "C" int drk_conv_fwd_x3(const float* h, const void* w3, const float* bias, const float* resid, float* out,
                               unsigned char* relu_mask, int B, int L, int taps, int dilation, int relu, void* stream) {
  DRK_SHAPE_GUARD("drk_conv_fwd_x3");
  if (!(h && w3 && out && taps >= 1 && (taps & 1) && dilation >= 0 && (const void*)h != (const void*)out))
    return drk_set_error_msg(DRK_EINVAL, "drk_conv_fwd_x3: bad arguments (h and out must be distinct buffers)");
  if (relu_mask && !relu) return drk_set_error_msg(DRK_EINVAL, "drk_conv_fwd_x3: a ReLU mask needs relu = 1");
  X3Args a;
  a.h = h; a.w3 = (const char*)w3; a.bias = bias; a.resid = resid; a.out = out; a.mask = relu_mask;
  a.M = (long)B * L; a.L = L; a.taps = taps; a.dil = dilation; a.relu = relu ? 1 : 0;
  if (a.M == 0) return DRK_OK;
  const int center = taps / 2;
  int reach = center;                      // taps whose shift is >= L read only padding: skipped
  if (dilation > 0 && (L - 1) / dilation < reach) reach = (L - 1) / dilation;
  a.first_tap = center - reach;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    static std::atomic<int> cu_count[DRK_MAX_DEVICES];
    if (dev >= 0 && dev < DRK_MAX_DEVICES) {
      cus = cu_count[dev].load(std::memory_order_relaxed);
      if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cu_count[dev].store(cus, std::memory_order_relaxed);
      }
    }
  }
  if (a.M >= (1L << 34)) return drk_set_error_msg(DRK_EINVAL, "drk_conv_fwd_x3: too many rows");
  const long t6 = (a.M + 95) / 96, t8 = (a.M + 127) / 128;
  const bool halo_form = center * dilation <= X3H_HALO_MAX;
  const bool halo64 = !halo_form && center * dilation <= 64;   // dilation 16: 64-row tiles around a 64-row halo
  // 128-row tiles where they take fewer row-rounds of the chip (few tiles: one round of 128 rows instead of two of 96)
  const bool big = halo_form && ((t8 + cus - 1) / cus) * 8 < ((t6 + cus - 1) / cus) * 6;
  const long ntiles = halo64 ? (a.M + 63) / 64 : big ? t8 : t6;
  const int per = (int)((ntiles + cus - 1) / cus);
  const int nwg = (int)((ntiles + per - 1) / per);
  if (halo64)
    hipLaunchKernelGGL((k_conv_x3_halo<4, 64>), dim3(nwg), dim3(512), 0, (hipStream_t)stream, a, per, (int)ntiles);
  else if (big)
    hipLaunchKernelGGL(k_conv_x3_halo<8>, dim3(nwg), dim3(512), 0, (hipStream_t)stream, a, per, (int)ntiles);
  else if (halo_form)
    hipLaunchKernelGGL(k_conv_x3_halo<6>, dim3(nwg), dim3(512), 0, (hipStream_t)stream, a, per, (int)ntiles);
  else
    hipLaunchKernelGGL(k_conv_x3, dim3(nwg), dim3(512), 0, (hipStream_t)stream, a, per, (int)ntiles);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
