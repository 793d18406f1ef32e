// k_group_bf16 — SEVERAL residual layers of the bf16 denoiser per launch for sequences longer than the fused sampler's
// 224 positions (BASELINE.json configs[4]: L = 1024), on the fused sampler's body: the residual stream of a 256-row
// image in registers, its LayerNorm output (bf16) in LDS, weights streamed from L2 in fragment order.  Between launches
// only the fp32 residual stream travels through HBM (1 KB per row and GROUP, against 1.5 KB per row and LAYER on the
// layer-per-launch path, drk_layer_bf16.hip, which this replaces where it applies).
//
// What lets a workgroup keep its rows for several layers (models/dnaconv.py:53-58: dilations 1,1,4,16,64, each repeated
// num_cnn_stacks times in a row; 88-101: the residual layers):
//   * stride mode (layers whose dilations are all multiples of s): the positions p = r (mod s) of one sequence are closed
//     under every shift of those layers, so image row k <-> position r + s k, a shift by d positions is a shift by d / s
//     image rows, rows that leave the image are the convolution's zero padding, and NO halo exists.  With s = 4 and
//     L = 1024 the twelve layers of dilation 4, 16, 64 are ONE launch of B x 4 workgroups (in-image dilations 1, 4, 16).
//   * halo mode (dilation-1 layers, s = 1): image row k <-> position j T - halo + k with halo = 4 x (layers in the
//     group) rows recomputed on either side; of the 16 NT-row image the middle T = 16 NT x 2 - 2 halo rows are written.
// Rows whose position lies outside [0, L) are never written to the image, which starts out all zero: that is the 'same'
// padding at the sequence ends, in both modes, for every layer of the group.
//
// Work split, lane <-> data maps, tap loops: drk_sample_bf16.hip / drk_fused_common.h (8 waves = 4 channel tiles x 2
// position groups; wave (ct, grp) owns row tiles grp, grp + 2, ... and channels 32 ct .. 32 ct + 31).
#define ZROW 303                 // image rows 0..287 (stride mode: 0..255), then all-zero rows (HPAD more in front)
#define DRK_LIVE_END 256
#include "drk_fused_common.h"

#define G_ROWS 288               // halo mode: up to 18 tiles of 16 rows
#define G_STRIDE_ROWS 256        // stride mode: 16 tiles, rows 256.. are padding (DRK_LIVE_END)
#define G_MAX_LAYERS 12
#define G_IMG_BYTES ((HPAD + ZROW + 1) * ROWB)                    // 87 040
#define G_OFF_PART G_IMG_BYTES                                    // float [288][12]: LayerNorm partials
#define G_OFF_VECS (G_OFF_PART + G_ROWS * PART_STRIDE * 4)        // float [n+1][3][128]: blocks first_layer .. first_layer + n
#define G_OFF_TB (G_OFF_VECS + (G_MAX_LAYERS + 1) * 3 * 128 * 4)  // float [n][128]
#define G_OFF_STEMW (G_OFF_TB + G_MAX_LAYERS * 128 * 4)           // float [9][5][128]: the stem's weights (token input only)
#define G_OFF_TOKS (G_OFF_STEMW + 9 * 5 * 128 * 4)                // int [288 + 8]: tokens of image rows -4 .. 291
#define G_LDS_BYTES (G_OFF_TOKS + (G_ROWS + 8) * 4)               // 151 200
#define G_STG_STRIDE 132                                          // floats per row of the head's fp32 staging tile [128][132]

struct GroupArgs {
  const char* wfrag;        // fragment stream at the group's first layer (fastpath.py::pack_stream)
  const float* vecs;        // vector blocks at the group's first layer: [n+1][3][128] (block 0: -, gamma, beta of its LayerNorm)
  const float* tbias;       // [n][128] at the group's first layer, shared by the batch
  const float* feat_in;     // [B][L][128] fp32 residual stream; NULL with x_tok
  const int64_t* x_tok;     // [B][L] tokens: the stem (dnaconv.py:80-86) runs here, k_stem_tok's arithmetic (halo mode only)
  const float* stem_w;      // [9][5][128]
  const float* stem_b;      // [128]
  float* feat_out;          // NULL: not wanted (the stack's last group, whose h_out is all the head reads)
  char* h_out;              // bf16 [B][L][128] of the outgoing stream (final_conv.0's operand), or NULL
  float* logits;            // [B][L][5]: the head (dnaconv.py:60-63, 103-104) runs here, drk_head_bf16's arithmetic; or NULL
  const float* head_w;      // [5][128]
  const float* head_b;      // [5]
  int B, L, n;
  int stride, halo, T, tiles_per_seq;
  unsigned long long dilpk[2];   // in-image dilations, 8 bits a layer
};

// epilogue_ln of the sampler on this file's LDS map; `vmask` bit t: the lane's position of tile t lies inside the sequence.
template <int NT, bool ACCZERO>
__device__ __forceinline__ void epilogue_ln_group(char* smem, f32x4 (&feat)[NT][2], const f32x4 (&acc)[NT][2], const float* blk,
                                                  const float* tbv_src, bool do_ln, unsigned vmask, int lane, int ct, int grp) {
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
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) feat[t][s] += relu4(acc[t][s]);
  }
  if (!do_ln) return;
  float* part = (float*)(smem + G_OFF_PART);
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
  __syncthreads();   // partials complete; every wave is past its MFMA reads of the old image
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
    if ((vmask >> t) & 1u) store_h8(smem, p, ch0, hv);
  }
  __syncthreads();
}

template <int NT, int GRP, bool HAS16>
__device__ __forceinline__ void group_body(const GroupArgs& a, char* smem, int tid, int wave, int lane) {
  const int ct = wave & 3;
  constexpr int grp = GRP;
  const int L = a.L;
  const int b = (int)blockIdx.x / a.tiles_per_seq;
  const int j = (int)blockIdx.x - b * a.tiles_per_seq;
  const int p0 = a.stride > 1 ? j : j * a.T - a.halo;       // position of image row 0
  const float* vecs = (const float*)(smem + G_OFF_VECS);
  const float* tbl = (const float*)(smem + G_OFF_TB);
  auto dil_of = [&](int layer) { return (int)((a.dilpk[layer >> 3] >> (8 * (layer & 7))) & 255ull); };
  const char* wuni = a.wfrag + ct * (2 * SUB_BYTES);
  const unsigned loff = (unsigned)lane * 16u;

  for (int i = tid; i < G_IMG_BYTES / 16; i += 512) ((f32x4*)smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < (a.n + 1) * 3 * 128 / 4; i += 512) ((f32x4*)(smem + G_OFF_VECS))[i] = ((const f32x4*)a.vecs)[i];
  for (int i = tid; i < a.n * 128 / 4; i += 512) ((f32x4*)(smem + G_OFF_TB))[i] = ((const f32x4*)a.tbias)[i];
  bf16v8 ring[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) ring[f] = wload(wuni + (f >> 1) * KSTEP_BYTES + (f & 1) * SUB_BYTES, loff);

  f32x4 feat[NT][2], acc[NT][2];
  unsigned vmask = 0;
  if (a.x_tok) {   // (uniform) tokens in: the stem's weights and this image's tokens go to LDS first
    for (int i = tid; i < 9 * 5 * 128 / 4; i += 512) ((f32x4*)(smem + G_OFF_STEMW))[i] = ((const f32x4*)a.stem_w)[i];
    int* toks = (int*)(smem + G_OFF_TOKS);
    if (tid < 32 * NT + 8) {
      const int p = p0 - 4 + tid;
      toks[tid] = ((unsigned)p < (unsigned)L) ? (int)a.x_tok[(long)b * L + p] : -1;
    }
    __syncthreads();
  }
  {   // the residual stream: this lane's 8 channels of its positions
    const int lo = opaque(lane);
    const int pn = (lo & 15) ^ ((lo & 8) >> 1);
    const int ch0 = 32 * ct + 4 * (lo >> 4);
    if (a.x_tok) {
      // k_stem_tok's sum (drk_cnn_f32.hip): one weight row per tap picked by the token, taps in order, then the bias
      const float* wl = (const float*)(smem + G_OFF_STEMW) + ch0;
      const int* toks = (const int*)(smem + G_OFF_TOKS);
      f32x4 sb[2];
      load_vec8(a.stem_b, ch0, sb);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int k = 16 * (grp + 2 * t) + pn;
        const bool ok = (unsigned)(p0 + k) < (unsigned)L;
        vmask |= ok ? (1u << t) : 0u;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int tk = toks[k + tap];
          if ((unsigned)tk < 5u) {
            s0 += *(const f32x4*)(wl + (tap * 5 + tk) * 128);
            s1 += *(const f32x4*)(wl + (tap * 5 + tk) * 128 + 16);
          }
        }
        s0 += sb[0];
        s1 += sb[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          feat[t][0][i] = ok ? fmaxf(s0[i], 0.f) : 0.f;
          feat[t][1][i] = ok ? fmaxf(s1[i], 0.f) : 0.f;
        }
      }
    } else {
      const float* f0 = a.feat_in + (long)b * L * 128 + ch0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p = p0 + a.stride * (16 * (grp + 2 * t) + pn);
        const bool ok = (unsigned)p < (unsigned)L;
        vmask |= ok ? (1u << t) : 0u;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        feat[t][0] = ok ? *(const f32x4*)(f0 + (long)p * 128) : z;
        feat[t][1] = ok ? *(const f32x4*)(f0 + (long)p * 128 + 16) : z;
      }
    }
  }
  __syncthreads();   // image zeroed, tables resident
  epilogue_ln_group<NT, true>(smem, feat, acc, vecs, tbl, true, vmask, lane, ct, grp);

  for (int layer = 0; layer < a.n; ++layer) {
    const int dil = dil_of(layer);
    {
      f32x4 cbl[2];
      load_vec8(vecs + 384 * (layer + 1), 32 * ct + 4 * (opaque(lane) >> 4), cbl);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t][0] = cbl[0];
        acc[t][1] = cbl[1];
      }
    }
    const char* wn = wuni + (long)(9 * layer + 1) * TAP_BYTES;
    const char* wafter = wuni + (long)(9 * layer + 9) * TAP_BYTES;   // the stream always has a tap behind a layer's ninth
    if (HAS16 && dil == 16) {
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
    TAP_PRIO(false);
    epilogue_ln_group<NT, false>(smem, feat, acc, vecs + 384 * (layer + 1), tbl + 128 * (layer + 1), layer + 1 < a.n, vmask, lane,
                                 ct, grp);
  }

  // ---- the residual stream goes back: image rows halo .. halo + T - 1 that lie inside the sequence ----
  {
    const int lo = opaque(lane);
    const int pn = (lo & 15) ^ ((lo & 8) >> 1);
    const int ch0 = 32 * ct + 4 * (lo >> 4);
    float* fo = a.feat_out + (long)b * L * 128 + ch0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int k = 16 * (grp + 2 * t) + pn;
      const int p = p0 + a.stride * k;
      if (a.feat_out && ((vmask >> t) & 1u) && (unsigned)(k - a.halo) < (unsigned)a.T) {
        *(f32x4*)(fo + (long)p * 128) = feat[t][0];
        *(f32x4*)(fo + (long)p * 128 + 16) = feat[t][1];
      }
    }
    if (a.h_out || a.logits) {   // bf16 of the stream into the image
      __syncthreads();   // the last layer did no LayerNorm: every wave is past its reads of the image
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if ((vmask >> t) & 1u) store_h8(smem, 16 * (grp + 2 * t) + pn, ch0, feat[t]);
      __syncthreads();
    }
    if (a.h_out) {   // whole 256-byte rows leave
      char* ho = a.h_out + (long)b * L * 256;
      for (int i = tid; i < a.T * 16; i += 512) {
        const int k = a.halo + (i >> 4);
        const int p = p0 + a.stride * k;
        if ((unsigned)p < (unsigned)L)
          *(f32x4*)(ho + (long)p * 256 + 16 * (i & 15)) = *(const f32x4*)(smem + OFF_HB + k * ROWB + 16 * (i & 15));
      }
    }
  }
  if (!a.logits) return;

  // ---- the head: final_conv.0 (1x1, its fragments are in the ring: the tap behind the last layer's ninth) + ReLU, then
  // final_conv.2 (128 -> 5) in fp32 with k_head_out's arithmetic on the y tile staged through LDS, 128 rows at a time
  {
#pragma clang fp contract(off)
    {
      f32x4 cbl[2];
      load_vec8(vecs + 384 * a.n + 128, 32 * ct + 4 * (opaque(lane) >> 4), cbl);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t][0] = cbl[0];
        acc[t][1] = cbl[1];
      }
      const char* hcur = pad_base(smem, lane);
      bf16v8 cur[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) cur[t] = *(const bf16v8*)(hcur + RI(GRP, t, 0));
      mma_tap_pad<NT, GRP, false>(hcur, hcur, wuni + (long)(9 * a.n) * TAP_BYTES, loff, ring, cur, acc);   // (the refill is unused)
    }
    const int lo = opaque(lane);
    const int pn = (lo & 15) ^ ((lo & 8) >> 1);
    const int ch0 = 32 * ct + 4 * (lo >> 4);
    float* const stg = (float*)(smem + OFF_HB - HPAD * ROWB);      // [128][132] floats = 67 584 B of the image area
    float* const ws = (float*)(smem + G_OFF_PART);                 // [5][128]
    float* lg = a.logits + (long)b * L * DRK_V;
    const int s8 = tid & 7;
#pragma unroll
    for (int half = 0; half < (2 * NT + 7) / 8; ++half) {   // 128 image rows = the wave's tiles 4 half .. 4 half + 3
      __syncthreads();   // every wave is past its reads of the image (half 0) / of the staged rows
      if (half == 0)
        for (int i = tid; i < DRK_V * DRK_C; i += 512) ws[i] = a.head_w[i];
#pragma unroll
      for (int t = 4 * half; t < (4 * half + 4 < NT ? 4 * half + 4 : NT); ++t) {
        const int rl = 16 * (grp + 2 * t) + pn - 128 * half;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          f32x4 yv;
#pragma unroll
          for (int i = 0; i < 4; ++i) yv[i] = fmaxf(acc[t][s][i], 0.f);
          *(f32x4*)(stg + rl * G_STG_STRIDE + ch0 + 16 * s) = yv;
        }
      }
      __syncthreads();
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        const int rowl = 64 * pass + (tid >> 3);
        const int k = 128 * half + rowl;
        const int p = p0 + a.stride * k;
        const bool ok = k < 32 * NT && (unsigned)p < (unsigned)L && (unsigned)(k - a.halo) < (unsigned)a.T;
        float part[DRK_V] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int ch = 4 * (s8 + 8 * c);
            const f32x4 av = *(const f32x4*)(stg + rowl * G_STG_STRIDE + ch);
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
        if (ok && s8 < DRK_V) {
          float o = part[0];
#pragma unroll
          for (int v = 1; v < DRK_V; ++v) o = (s8 == v) ? part[v] : o;
          lg[(long)p * DRK_V + s8] = o + a.head_b[s8];
        }
      }
    }
  }
}

template <int NT, bool HAS16>
__global__ void __launch_bounds__(512, 2) k_group_bf16(const GroupArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4)
    group_body<NT, 0, HAS16>(a, smem, tid, wave, lane);
  else
    group_body<NT, 1, HAS16>(a, smem, tid, wave, lane);
}

// Rows of feat_out one workgroup writes in halo mode (stride 1), 0 when the group does not fit an image of `tiles16`
// 16-row tiles; workgroups per sequence = ceil(L / that).
extern "C" long drk_layer_group_bf16_rows(int n_group, int tiles16) {
  if (n_group < 1 || n_group > G_MAX_LAYERS || (tiles16 != 14 && tiles16 != 16 && tiles16 != 18)) return 0;
  const long T = 16L * tiles16 - 8L * n_group;
  return T >= 16 ? T : 0;
}

// Layers first_layer .. first_layer + n_group - 1 of the residual stack in one launch (see the head of this file).
//   stride > 1: every dilation a multiple of `stride`, dilation / stride in {1..4, 16}, ceil(L / stride) <= 256; tiles16 = 16;
//               feat_out may be feat_in.
//   stride = 1: every dilation in 1..4 (halo = 4 x their sum), tiles16 = 14, 16 or 18 (image rows / 16), feat_out != feat_in.
// wstream / vecs / tbias: the WHOLE model's fragment stream, vector table and time-bias table [n_layers][128] (one time
// for the batch), as drk_sample_bf16 takes them; dilations: the group's n_group TRUE dilations.
// In: feat_in, or (first_layer = 0, stride = 1) the tokens x_tok with the stem's stem_w_t [9][5][128] / stem_b [128]: the
// stem runs in the launch.  Out, any of: feat_out (fp32 stream); h_out (its bf16 rows, drk_head_bf16's operand); logits
// [B*L][5] with head_w [5][128] / head_b [5]: the head runs in the launch (the caller says so only for the group that
// ends the stack: the tap behind the group's last layer is then final_conv.0).
extern "C" int drk_layer_group_bf16(const float* feat_in, const int64_t* x_tok, const float* stem_w_t, const float* stem_b,
                                    float* feat_out, void* h_out, float* logits, const float* head_w, const float* head_b,
                                    const void* wstream, const float* vecs, const float* tbias, int B, int L, int first_layer,
                                    int n_group, int stride, int tiles16, const int* dilations, void* stream) {
  DRK_SHAPE_GUARD("drk_layer_group_bf16");
  if ((!feat_in == !x_tok) || (!feat_out && !h_out && !logits) || !wstream || !vecs || !tbias || !dilations)
    return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: null pointer (one of feat_in / x_tok, at least one output)");
  if (x_tok && (!stem_w_t || !stem_b || first_layer != 0 || stride != 1))
    return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: tokens come with stem_w_t, stem_b, first_layer = 0, stride = 1");
  if (logits && (!head_w || !head_b))
    return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: logits come with head_w, head_b");
  if (((uintptr_t)feat_in | (uintptr_t)feat_out | (uintptr_t)h_out | (uintptr_t)wstream | (uintptr_t)vecs | (uintptr_t)tbias |
       (uintptr_t)stem_w_t | (uintptr_t)stem_b | (uintptr_t)head_w) & 15)
    return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: 16-byte aligned pointers");
  if (first_layer < 0 || n_group < 1 || n_group > G_MAX_LAYERS || stride < 1 ||
      (tiles16 != 14 && tiles16 != 16 && tiles16 != 18))
    return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: 1..12 layers, stride >= 1, tiles16 = 14, 16 or 18");
  GroupArgs a;
  a.dilpk[0] = a.dilpk[1] = 0;
  bool has16 = false;
  int reach = 0;
  for (int i = 0; i < n_group; ++i) {
    const int d = dilations[i];
    if (d < 1 || d % stride) return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: every dilation a multiple of stride");
    const int di = d / stride;
    if (!((di >= 1 && di <= 4) || (di == 16 && stride > 1)))
      return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: dilation / stride in {1..4} (or 16 when stride > 1)");
    has16 = has16 || di == 16;
    reach += 4 * di;
    a.dilpk[i >> 3] |= (unsigned long long)di << (8 * (i & 7));
  }
  if (stride > 1) {
    if (tiles16 != 16 || ((long)L + stride - 1) / stride > G_STRIDE_ROWS)
      return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: stride > 1 takes tiles16 = 16 and ceil(L / stride) <= 256");
    a.halo = 0; a.T = G_STRIDE_ROWS; a.tiles_per_seq = stride;
  } else {
    a.halo = reach; a.T = 16 * tiles16 - 2 * reach;
    if (a.T < 16) return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: the group's halo leaves no rows to write");
    if (feat_in && feat_in == feat_out) return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: stride 1 reads its neighbours' rows: feat_out != feat_in");
    a.tiles_per_seq = (L + a.T - 1) / a.T;
  }
  const long grid = (long)B * a.tiles_per_seq;
  if (grid >= (1L << 31)) return drk_set_error_msg(DRK_EINVAL, "drk_layer_group_bf16: too many workgroups");
  a.wfrag = (const char*)wstream + (long)first_layer * 9 * TAP_BYTES;
  a.vecs = vecs + (long)first_layer * 384;
  a.tbias = tbias + (long)first_layer * 128;
  a.feat_in = feat_in; a.feat_out = feat_out; a.h_out = (char*)h_out;
  a.x_tok = x_tok; a.stem_w = stem_w_t; a.stem_b = stem_b; a.logits = logits; a.head_w = head_w; a.head_b = head_b;
  a.B = B; a.L = L; a.n = n_group; a.stride = stride;
  const int which = has16 ? 2 : (tiles16 == 18 ? 3 : tiles16 == 16 ? 1 : 0);
  const void* kern = which == 2 ? (const void*)k_group_bf16<8, true> : which == 3 ? (const void*)k_group_bf16<9, false>
                   : which == 1 ? (const void*)k_group_bf16<8, false> : (const void*)k_group_bf16<7, false>;
  static std::atomic<unsigned> lds_set[4][DRK_MAX_DEVICES];
  {
    int devid = 0;
    hipError_t e = hipGetDevice(&devid);
    if (e != hipSuccess) return drk_set_error(e, "drk_layer_group_bf16(hipGetDevice)");
    if (devid < 0 || devid >= DRK_MAX_DEVICES || !lds_set[which][devid].load(std::memory_order_acquire)) {
      e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES);   // not an allocation
      if (e != hipSuccess) return drk_set_error(e, "drk_layer_group_bf16(LDS size)");
      if (devid >= 0 && devid < DRK_MAX_DEVICES) lds_set[which][devid].store(1u, std::memory_order_release);
    }
  }
  if (which == 2)
    hipLaunchKernelGGL((k_group_bf16<8, true>), dim3((unsigned)grid), dim3(512), G_LDS_BYTES, (hipStream_t)stream, a);
  else if (which == 3)
    hipLaunchKernelGGL((k_group_bf16<9, false>), dim3((unsigned)grid), dim3(512), G_LDS_BYTES, (hipStream_t)stream, a);
  else if (which == 1)
    hipLaunchKernelGGL((k_group_bf16<8, false>), dim3((unsigned)grid), dim3(512), G_LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((k_group_bf16<7, false>), dim3((unsigned)grid), dim3(512), G_LDS_BYTES, (hipStream_t)stream, a);
  DRK_CHECK_LAUNCH();
  return DRK_OK;
}
