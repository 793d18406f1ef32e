// bf16-MFMA backward kernels of the CNN denoiser, for the truncated reward backpropagation of
// Diffusion._sample_finetune_gradient (diffusion_gosai_update.py:552-630) at BASELINE.json configs[2] speed.  The reference
// gets these from torch autograd over CNNModel.forward (models/dnaconv.py:88-101); its own GPU convolutions ran in
// TF32 (cuDNN default under autocast(float32), diffusion_gosai_update.py:270), so bf16 operands with fp32
// accumulation, fp32 master weights, an fp32 residual-gradient stream and fp32 LayerNorm statistics are a like-for-like
// trade.  The exact-fp32 kernels (drk_train_f32.hip) stay the parity path.
//
//   * input gradient of a conv  = drk_layer_bf16 (drk_layer_bf16.hip) on dc and the transposed, tap-flipped stream;
//   * k_gate_bf16               : dc = bf16(g * relu-mask) + its column sums (the conv bias gradient);
//   * k_wgrad_bf16              : dW[t][ci][co] = sum_rows h[row + off_t][ci] * dc[row][co], split-K over row chunks,
//                                 DETERMINISTIC (per-chunk slabs + ordered reduction, no atomics).
//
// k_wgrad_bf16.  A GEMM whose K dimension is the ROW index of two row-major [rows][128] bf16 tensors, so both MFMA
// operands need 8 consecutive ROWS of one channel per lane: the transpose of what memory holds.  Tiles of 64 rows of h
// (shifted by the tap, rows whose shifted position leaves [0, L) zeroed: 'same' padding and sequence ends) and of dc
// are staged through registers into LDS as plain 256-byte rows with the 16-byte-chunk XOR of cdna_hip_programming.md
// T10 image (b), and read back with ds_read_b64_tr_b16 (the hardware transpose read): two reads = one operand fragment,
// conflict-free for the 16x16x32 operand.  v_mfma_f32_16x16x32_bf16, A = h^T [16 ci][32 rows], B = dc [32 rows][16 co];
// workgroup = (row chunk, tap), 4 waves as 2 (ci halves) x 2 (co halves), 4 x 4 accumulator tiles per wave; two LDS
// buffers, the next tile's global loads in flight under this tile's 32 MFMAs per wave, one barrier per tile.
#include "drk_common.h"
#include "../../include/drakes_hip.h"

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define BARGCHECK(cond, name) \
  if (!(cond)) return drk_set_error_msg(DRK_EINVAL, name ": bad arguments")

#define WB_TILE 64                       // rows per LDS tile = 2 k-steps of 32
#define WB_IMG (WB_TILE * 256)           // bytes of one image (64 rows x 128 bf16)
#define WB_SLAB (DRK_C * DRK_C)          // floats per (chunk, tap) slab
#define WB_MAX_SLABS 252                 // one round of workgroups on the 256 CUs; bounds the workspace at 16.5 MB
#define WB_MAX_TAPS 9

// byte offset of 16-byte chunk `ch` of row `row` in an image of plain 256-byte rows (T10 image (b)): serves row-wise
// 16-byte stores and the transposed reads without bank conflicts
__device__ __forceinline__ int wb_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__device__ __forceinline__ bf16v8 tr_frag(const char* p0, const char* p1) {
  typedef s16x4 __attribute__((address_space(3))) * lds4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)p1);
  return __builtin_bit_cast(bf16v8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// MULTI: the rows are the concatenation of n segments of M rows each (the K recorded steps of one fine-tune chain, each
// with its own h and dc buffers: hs[seg], ds[seg] are device arrays of base pointers); a 64-row tile never straddles two
// segments (tiles_per_seg = ceil(M / 64) tiles each), a workgroup owns tiles_per_wg consecutive tiles of the sequence.
template <bool MULTI>
__global__ void __launch_bounds__(256, 2) k_wgrad_bf16(const char* __restrict__ h, const char* __restrict__ dc,
                                                       const char* const* __restrict__ hs, const char* const* __restrict__ ds,
                                                       float* __restrict__ slabs, long M, int L, int taps, int dil,
                                                       int tiles_per_seg, int tiles_per_wg, int total_tiles) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * WB_IMG];   // [buffer][h image, dc image]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;                 // this wave: ci 64 wr .. +63, co 64 wc .. +63
  const int t = blockIdx.y;
  const int off = (t - taps / 2) * dil;
  const int T0 = blockIdx.x * tiles_per_wg;
  const int ntiles = (T0 + tiles_per_wg < total_tiles ? T0 + tiles_per_wg : total_tiles) - T0;

  // staging role: rows srow + 16 i (i < 4) of a tile, 16-byte chunk scol
  const int srow = tid >> 4, scol = tid & 15;
  int pos[4], pos0[4];
  int lt = T0 % tiles_per_seg, seg = T0 / tiles_per_seg;    // local tile in its segment, segment
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pos0[i] = (srow + 16 * i) % L;
    pos[i] = (int)(((long)lt * WB_TILE + srow + 16 * i) % L);
  }
  const int pstep = WB_TILE % L;
  f32x4 sh[4], sd[4];
  auto fetch = [&]() {          // the next tile of the sequence (lt, seg), then advance
    const char* hb = MULTI ? hs[seg] : h;
    const char* db = MULTI ? ds[seg] : dc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long r = (long)lt * WB_TILE + srow + 16 * i;
      const bool vh = r < M && (unsigned)(pos[i] + off) < (unsigned)L;
      sh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      sd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (vh) {   // a dead h row makes the product zero whatever dc holds: skip both loads
        sh[i] = *(const f32x4*)(hb + (r + off) * 256 + 16 * scol);
        sd[i] = *(const f32x4*)(db + r * 256 + 16 * scol);
      }
      pos[i] += pstep;
      if (pos[i] >= L) pos[i] -= L;
    }
    if (++lt == tiles_per_seg) {
      lt = 0;
      ++seg;
#pragma unroll
      for (int i = 0; i < 4; ++i) pos[i] = pos0[i];
    }
  };
  auto park = [&](int buf) {
    char* base = smem + buf * (2 * WB_IMG);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = wb_off(srow + 16 * i, scol);
      *(f32x4*)(base + o) = sh[i];
      *(f32x4*)(base + WB_IMG + o) = sd[i];
    }
  };

  // transposed-read addresses (k-step 0, buffer 0): lane 16 kg + 4 q + p supplies row 8 kg + q (+4 for the second
  // read), columns 16 T + 4 p .. + 3 of tile column block T, i.e. chunk 2 T + (p >> 1), byte 8 (p & 1)
  const int kg = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  int aoff[4][2], boff[4][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 8 * kg + q + 4 * e;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      aoff[m][e] = wb_off(row, 2 * (4 * wr + m) + (p >> 1)) + 8 * (p & 1);
      boff[m][e] = WB_IMG + wb_off(row, 2 * (4 * wc + m) + (p >> 1)) + 8 * (p & 1);
    }
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch();
  park(0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool more = tile + 1 < ntiles;
    if (more) fetch();                                // in flight under this tile's MFMAs
    const char* base = smem + (tile & 1) * (2 * WB_IMG);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // a k-step is 32 rows further: + 32 * 256 bytes (the swizzle depends on row mod 16 only)
      bf16v8 af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        af[m] = tr_frag(base + ks * 8192 + aoff[m][0], base + ks * 8192 + aoff[m][1]);
        bf[m] = tr_frag(base + ks * 8192 + boff[m][0], base + ks * 8192 + boff[m][1]);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
    if (more) park((tile + 1) & 1);                   // its last readers passed the previous barrier
    __syncthreads();
  }

  // D[row 4 g + i][col n] of tile (m, n): ci = 64 wr + 16 m + 4 g + i, co = 64 wc + 16 n + (lane & 15)
  float* dst = slabs + ((long)blockIdx.x * taps + t) * WB_SLAB;
  const int g = lane >> 4, cn = lane & 15;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(64 * wr + 16 * m + 4 * g + i) * DRK_C + 64 * wc + 16 * n + cn] = acc[m][n][i];
}

// dc = bf16(g * gate), gate = relu mask (uint8, written by drk_layer_bf16 / drk_conv_fwd_f32) or (y > 0) for a saved
// ReLU output y; column sums of the fp32 products (the bias gradient) as per-workgroup partials in `part` (slab
// blockIdx.x of 128 floats; reduced in order by the caller).  32 lanes per row, 4 channels per lane.
__global__ void __launch_bounds__(256) k_gate_bf16(const float* __restrict__ g, const unsigned char* __restrict__ mask,
                                                   const float* __restrict__ y, char* __restrict__ out,
                                                   float* __restrict__ part, long M) {
  __shared__ float red[8][DRK_C];
  const int lr = threadIdx.x & 31, sub = threadIdx.x >> 5;
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};
  for (long row = (long)blockIdx.x * 8 + sub; row < M; row += (long)gridDim.x * 8) {
    f32x4 v = *(const f32x4*)(g + row * DRK_C + 4 * lr);
    if (mask) {
      const unsigned m = *(const unsigned*)(mask + row * DRK_C + 4 * lr);
      v = f32x4{(m & 0xffu) ? v[0] : 0.f, (m & 0xff00u) ? v[1] : 0.f, (m & 0xff0000u) ? v[2] : 0.f, (m & 0xff000000u) ? v[3] : 0.f};
    } else {
      const f32x4 yy = *(const f32x4*)(y + row * DRK_C + 4 * lr);
      v = f32x4{yy[0] > 0.f ? v[0] : 0.f, yy[1] > 0.f ? v[1] : 0.f, yy[2] > 0.f ? v[2] : 0.f, yy[3] > 0.f ? v[3] : 0.f};
    }
    cs += v;
    bf16v4 pk;
    pk[0] = (__bf16)v[0]; pk[1] = (__bf16)v[1]; pk[2] = (__bf16)v[2]; pk[3] = (__bf16)v[3];
    *(bf16v4*)(out + row * 256 + 8 * lr) = pk;
  }
  if (!part) return;
  *(f32x4*)(&red[sub][4 * lr]) = cs;
  __syncthreads();
  if (threadIdx.x < DRK_C) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
    part[(long)blockIdx.x * DRK_C + threadIdx.x] = s;
  }
}

// dc = bf16((dlogits x final_conv.2) * (y > 0)): the 128 -> 5 output conv's input gradient through final_conv.0's ReLU
// (models/dnaconv.py:61-63), in one pass over y, and per-workgroup partial sums [HB_SLAB] = dw2 [5][128] (sum_rows
// dlogits[v] * y[c]), db0 [128] (column sums of the fp32 gated gradient = final_conv.0's bias gradient), db2 [5] (+ 3 pad).
// 32 lanes per row (4 channels each), 8 rows per pass.
#define HB_ROWS 128
#define HB_SLAB (6 * DRK_C + 8)
__global__ void __launch_bounds__(256) k_head_bwd_bf16(const float* __restrict__ dlogits, const float* __restrict__ y,
                                                       const float* __restrict__ w, char* __restrict__ out,
                                                       float* __restrict__ partials, long part_stride, long M) {
  __shared__ __attribute__((aligned(16))) float red[8][HB_SLAB];
  const int lr = threadIdx.x & 31, sub = threadIdx.x >> 5;
  f32x4 wv[DRK_V];
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) wv[v] = *(const f32x4*)(w + v * DRK_C + 4 * lr);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 aw[DRK_V] = {zero4, zero4, zero4, zero4, zero4}, a0 = zero4;
  float ab[DRK_V] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const long rb = (long)blockIdx.x * HB_ROWS + sub;
#pragma unroll 1
  for (int k0 = 0; k0 < HB_ROWS / 8; k0 += 4) {
    f32x4 yy[4];
    float dl[4][DRK_V];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long row = rb + 8 * (k0 + j);
      const long r = row < M ? row : M - 1;
      yy[j] = *(const f32x4*)(y + r * DRK_C + 4 * lr);
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) dl[j][v] = dlogits[r * DRK_V + v];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long row = rb + 8 * (k0 + j);
      if (row >= M) continue;               // (uniform over the 32 lanes of a row)
      f32x4 o = zero4;
#pragma unroll
      for (int v = 0; v < DRK_V; ++v) {
        o += wv[v] * dl[j][v];
        aw[v] += yy[j] * dl[j][v];
        ab[v] += dl[j][v];
      }
      const f32x4 g = {yy[j][0] > 0.f ? o[0] : 0.f, yy[j][1] > 0.f ? o[1] : 0.f, yy[j][2] > 0.f ? o[2] : 0.f, yy[j][3] > 0.f ? o[3] : 0.f};
      a0 += g;
      bf16v4 pk;
      pk[0] = (__bf16)g[0]; pk[1] = (__bf16)g[1]; pk[2] = (__bf16)g[2]; pk[3] = (__bf16)g[3];
      *(bf16v4*)(out + row * 256 + 8 * lr) = pk;
    }
  }
  if (!partials) return;
#pragma unroll
  for (int v = 0; v < DRK_V; ++v) *(f32x4*)(&red[sub][v * DRK_C + 4 * lr]) = aw[v];
  *(f32x4*)(&red[sub][5 * DRK_C + 4 * lr]) = a0;
  if (lr < 8) red[sub][6 * DRK_C + lr] = lr < DRK_V ? (lr == 0 ? ab[0] : lr == 1 ? ab[1] : lr == 2 ? ab[2] : lr == 3 ? ab[3] : ab[4]) : 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < HB_SLAB; i += 256) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][i];
    partials[(long)blockIdx.x * part_stride + i] = s;
  }
}

// ------------------------------------------------------------------------------------ C ABI
static void wb_plan(long tiles, int taps, int max_slabs, int* tiles_per_wg, int* chunks) {
  long maxx = max_slabs / taps;
  if (maxx < 1) maxx = 1;
  const long tpw = (tiles + maxx - 1) / maxx;
  *tiles_per_wg = (int)tpw;
  *chunks = (int)((tiles + tpw - 1) / tpw);
}

extern "C" long drk_conv_wgrad_bf16_workspace_bytes(int B, int L, int taps) {
  if (B <= 0 || L <= 0 || taps < 1 || taps > WB_MAX_TAPS) return 0;
  int tpw, chunks;
  wb_plan(((long)B * L + WB_TILE - 1) / WB_TILE, taps, WB_MAX_SLABS, &tpw, &chunks);
  return (long)chunks * taps * WB_SLAB * (long)sizeof(float);
}

extern "C" int drk_conv_wgrad_bf16(const void* h, const void* dc, float* dwt, int B, int L, int taps, int dilation,
                                   void* workspace, long workspace_bytes, void* stream) {
  DRK_SHAPE_GUARD("drk_conv_wgrad_bf16");
  BARGCHECK(h && dc && dwt && taps >= 1 && taps <= WB_MAX_TAPS && (taps & 1) && dilation >= 0, "drk_conv_wgrad_bf16");
  BARGCHECK(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= drk_conv_wgrad_bf16_workspace_bytes(B, L, taps),
            "drk_conv_wgrad_bf16 (workspace: 16-byte aligned, drk_conv_wgrad_bf16_workspace_bytes(B, L, taps) bytes)");
  const long M = (long)B * L;
  const long tiles = (M + WB_TILE - 1) / WB_TILE;
  BARGCHECK(tiles < (1L << 30), "drk_conv_wgrad_bf16 (too many rows)");
  int tpw, chunks;
  wb_plan(tiles, taps, WB_MAX_SLABS, &tpw, &chunks);
  hipLaunchKernelGGL((k_wgrad_bf16<false>), dim3((unsigned)chunks, (unsigned)taps), dim3(256), 0, (hipStream_t)stream,
                     (const char*)h, (const char*)dc, nullptr, nullptr, (float*)workspace, M, L, taps, dilation, (int)tiles, tpw,
                     (int)tiles);
  DRK_CHECK_LAUNCH();
  return drk_slab_reduce_launch((const float*)workspace, chunks, (long)taps * WB_SLAB, dwt, (long)taps * WB_SLAB,
                                (hipStream_t)stream);
}

// Many recorded (h, dc) pairs, one launch: two workgroups per CU (the launch is long enough to fill them)
#define WB_MULTI_MAX_SLABS 504
extern "C" long drk_conv_wgrad_bf16_multi_workspace_bytes(int n_seg, int B, int L, int taps) {
  if (n_seg <= 0 || B <= 0 || L <= 0 || taps < 1 || taps > WB_MAX_TAPS) return 0;
  int tpw, chunks;
  wb_plan((((long)B * L + WB_TILE - 1) / WB_TILE) * n_seg, taps, WB_MULTI_MAX_SLABS, &tpw, &chunks);
  return (long)chunks * taps * WB_SLAB * (long)sizeof(float);
}

extern "C" int drk_conv_wgrad_bf16_multi(const void* const* h_ptrs, const void* const* dc_ptrs, int n_seg, float* dwt, int B,
                                         int L, int taps, int dilation, void* workspace, long workspace_bytes, void* stream) {
  DRK_SHAPE_GUARD("drk_conv_wgrad_bf16_multi");
  if (n_seg == 0) return DRK_OK;
  BARGCHECK(n_seg > 0 && h_ptrs && dc_ptrs && dwt && taps >= 1 && taps <= WB_MAX_TAPS && (taps & 1) && dilation >= 0,
            "drk_conv_wgrad_bf16_multi");
  BARGCHECK(workspace && ((uintptr_t)workspace & 15) == 0 &&
                workspace_bytes >= drk_conv_wgrad_bf16_multi_workspace_bytes(n_seg, B, L, taps),
            "drk_conv_wgrad_bf16_multi (workspace: 16-byte aligned, drk_conv_wgrad_bf16_multi_workspace_bytes(...) bytes)");
  const long M = (long)B * L;
  const long tps = (M + WB_TILE - 1) / WB_TILE, tiles = tps * n_seg;
  BARGCHECK(tiles < (1L << 30), "drk_conv_wgrad_bf16_multi (too many rows)");
  int tpw, chunks;
  wb_plan(tiles, taps, WB_MULTI_MAX_SLABS, &tpw, &chunks);
  hipLaunchKernelGGL((k_wgrad_bf16<true>), dim3((unsigned)chunks, (unsigned)taps), dim3(256), 0, (hipStream_t)stream, nullptr,
                     nullptr, (const char* const*)h_ptrs, (const char* const*)dc_ptrs, (float*)workspace, M, L, taps, dilation,
                     (int)tps, tpw, (int)tiles);
  DRK_CHECK_LAUNCH();
  return drk_slab_reduce_launch((const float*)workspace, chunks, (long)taps * WB_SLAB, dwt, (long)taps * WB_SLAB,
                                (hipStream_t)stream);
}

extern "C" long drk_head_bwd_bf16_slab_floats(void) { return HB_SLAB; }

extern "C" int drk_head_bwd_bf16(const float* dlogits, const float* y, const float* w, void* dc, float* partials,
                                 long part_stride, int B, int L, void* stream) {
  DRK_SHAPE_GUARD("drk_head_bwd_bf16");
  BARGCHECK(dlogits && y && w && dc, "drk_head_bwd_bf16");
  BARGCHECK(!partials || part_stride >= HB_SLAB, "drk_head_bwd_bf16 (partials: slabs of part_stride >= 776 floats)");
  const long M = (long)B * L;
  hipLaunchKernelGGL(k_head_bwd_bf16, dim3((unsigned)((M + HB_ROWS - 1) / HB_ROWS)), dim3(256), 0, (hipStream_t)stream, dlogits,
                     y, w, (char*)dc, partials, part_stride, M);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}

#define GATE_MAX_WG 256
extern "C" long drk_gate_bf16_workspace_bytes(void) { return (long)GATE_MAX_WG * DRK_C * (long)sizeof(float); }

extern "C" int drk_gate_bf16(const float* g, const unsigned char* relu_mask, const float* y, void* out, float* colsum, int B,
                             int L, void* workspace, long workspace_bytes, void* stream) {
  DRK_SHAPE_GUARD("drk_gate_bf16");
  BARGCHECK(g && out && (!relu_mask != !y), "drk_gate_bf16 (exactly one of relu_mask / y)");
  BARGCHECK(!colsum || (workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= drk_gate_bf16_workspace_bytes()),
            "drk_gate_bf16 (column sums need a workspace of drk_gate_bf16_workspace_bytes() bytes)");
  const long M = (long)B * L;
  long grid = (M + 63) / 64;
  if (grid > GATE_MAX_WG) grid = GATE_MAX_WG;
  hipLaunchKernelGGL(k_gate_bf16, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, g, relu_mask, y, (char*)out,
                     colsum ? (float*)workspace : nullptr, M);
  DRK_CHECK_LAUNCH();
  if (colsum) return drk_slab_reduce_launch((const float*)workspace, (int)grid, (long)DRK_C, colsum, (long)DRK_C, (hipStream_t)stream);
  return DRK_OK;
}
