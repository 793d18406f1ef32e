// One C-ABI call per differentiable denoiser forward / backward of the bf16 path.
//
// The K Gumbel-softmax steps of Diffusion._sample_finetune_gradient (diffusion_gosai_update.py:589-630) run the CNN
// denoiser (models/dnaconv.py:74-110) K times forward and K times backward at fine-tune batch sizes (32-128 sequences:
// 50-200 workgroups per launch), where a launch lasts ~20 us and a Python-side launch sequence (one ctypes call, four
// allocations and a dozen pointer checks per layer) costs more host time than the kernels take on the GPU.  These two
// entry points are that launch sequence in native code: the caller hands over ONE block of memory per kind of saved
// tensor, the entry point walks the layers.  Nothing here allocates or synchronises; every launch goes through the
// per-kernel entry points of include/drakes_hip.h, so arguments are checked as they are there.
#include "drk_common.h"
#include "../../include/drakes_hip.h"

#define CARGCHECK(cond, name) \
  if (!(cond)) return drk_set_error_msg(DRK_EINVAL, name ": bad arguments")
#define TRY(call)                 \
  do {                            \
    const int rc__ = (call);      \
    if (rc__ != DRK_OK) return rc__; \
  } while (0)

#define CH_TAP_BYTES 32768L      // one tap of the fragment stream (drakes_amd/fastpath.py::_frag_tap)
#define CH_SLOT (4 * DRK_C)      // floats one drk_layer_bwd_bf16 launch leaves per workgroup

extern "C" int drk_denoiser_is_fused(int L, int n_layers, int flags) {
  return L >= 1 && L <= 224 && n_layers >= 1 && n_layers <= 20 && !(flags & DRK_DENOISER_PER_LAYER);
}

extern "C" long drk_denoiser_feats_floats(int B, int L, int n_layers, int flags) {
  if (B <= 0 || L <= 0 || n_layers < 1) return 0;
  const long M = (long)B * L;
  return drk_denoiser_is_fused(L, n_layers, flags) ? M * DRK_C + drk_train_fused_featn_floats(B, n_layers) : (n_layers + 1L) * M * DRK_C;
}

extern "C" long drk_denoiser_masks_bytes(int B, int L, int n_layers, int flags) {
  if (B <= 0 || L <= 0 || n_layers < 1) return 0;
  return drk_denoiser_is_fused(L, n_layers, flags) ? drk_train_fused_maskp_bytes(B, n_layers) : (long)n_layers * B * L * DRK_C;
}

extern "C" int drk_denoiser_fwd_bf16(const int64_t* x, const float* x_soft, const float* stem_w_t, const float* stem_b,
                                     const void* wstream, const float* vecs, const float* tbias, const float* head2_w,
                                     const float* head2_b, float* feats, void* hs, unsigned char* masks, float* means,
                                     float* rstds, float* y, float* logits, int B, int L, int n_layers, const int* dilations,
                                     int flags, void* stream) {
  DRK_SHAPE_GUARD("drk_denoiser_fwd_bf16");
  CARGCHECK((x || x_soft) && stem_w_t && stem_b && wstream && vecs && tbias && head2_w && head2_b && feats && hs && masks &&
                means && rstds && y && logits && n_layers >= 1 && dilations,
            "drk_denoiser_fwd_bf16");
  const long M = (long)B * L, n = n_layers;
  const char* ws = (const char*)wstream;
  char* h = (char*)hs;
  CARGCHECK(!(flags & ~DRK_DENOISER_PER_LAYER), "drk_denoiser_fwd_bf16 (unknown flag)");
  TRY(drk_stem_fwd(x, x_soft, stem_w_t, stem_b, feats, B, L, stream));
  if (drk_denoiser_is_fused(L, n_layers, flags)) {
    // one sequence per workgroup, all residual layers and final_conv.0 in ONE launch (drk_train_fused_bf16.hip); feats =
    // feat0 then the lane-order snapshots, masks = the packed gates
    TRY(drk_train_fwd_fused_bf16(wstream, vecs, tbias, feats, feats + M * DRK_C, hs, masks, means, rstds, y, B, L, n_layers,
                                 dilations, stream));
    return drk_head_out_fwd(y, head2_w, head2_b, logits, B, L, stream);
  }
  TRY(drk_ln_fwd_bf16(feats, tbias, 0, vecs + DRK_C, vecs + 2 * DRK_C, h, means, rstds, B, L, stream));
  for (long i = 0; i < n; ++i) {
    const bool last = i + 1 == n;
    const float* blk = vecs + 3 * (i + 1) * DRK_C;     // conv bias_i, gamma_{i+1}, beta_{i+1}
    TRY(drk_layer_bf16(h + i * M * 256, feats + i * M * DRK_C, feats + (i + 1) * M * DRK_C, h + (i + 1) * M * 256,
                       ws + 9 * i * CH_TAP_BYTES, blk, last ? nullptr : tbias + (i + 1) * DRK_C, 0, last ? nullptr : blk + DRK_C,
                       last ? nullptr : blk + 2 * DRK_C, B, L, 9, dilations[i], last ? 0 : 1, 1, masks + i * M * DRK_C,
                       last ? nullptr : means + (i + 1) * M, last ? nullptr : rstds + (i + 1) * M, stream));
  }
  TRY(drk_layer_bf16(h + n * M * 256, nullptr, y, nullptr, ws + 9 * n * CH_TAP_BYTES, vecs + (3 * n + 1) * DRK_C, nullptr, 0,
                     nullptr, nullptr, B, L, 1, 0, 0, 1, nullptr, nullptr, nullptr, stream));
  return drk_head_out_fwd(y, head2_w, head2_b, logits, B, L, stream);
}

// Slabs of the partials workspace: the launches' own (two per sequence on the fused path, one per 128 rows otherwise)
// plus room for the first stage of the ordered two-stage reduction (one slab per 16).
extern "C" long drk_denoiser_bwd_bf16_slabs(int B, int L) {
  if (B <= 0 || L <= 0) return 0;
  const long a = 2L * B, b = ((long)B * L + 127) / 128;
  const long m = a > b ? a : b;
  return m + (m + 15) / 16 + 1;
}

extern "C" long drk_denoiser_bwd_bf16_slab_floats(int n_layers) {
  return n_layers < 1 ? 0 : (long)(n_layers + 1) * CH_SLOT + drk_head_bwd_bf16_slab_floats();
}

extern "C" int drk_denoiser_bwd_bf16(const float* dlogits, const float* y, const float* head2_w, const void* wstream_t,
                                     const float* vecs, const float* tbias, const float* feats, const unsigned char* masks,
                                     const float* means, const float* rstds, void* dcs, int dcs_slots, float* gr,
                                     float* partials, float* acc, const int64_t* x, const float* x_soft,
                                     const float* stem_w_t, float* stem_dw, float* stem_db, float* dx, int B, int L,
                                     int n_layers, const int* dilations, int flags, void* stream) {
  DRK_SHAPE_GUARD("drk_denoiser_bwd_bf16");
  CARGCHECK(!(flags & ~DRK_DENOISER_PER_LAYER), "drk_denoiser_bwd_bf16 (unknown flag)");
  CARGCHECK(dlogits && y && head2_w && wstream_t && vecs && tbias && feats && masks && means && rstds && dcs && gr &&
                n_layers >= 1 && dilations && (x || x_soft) && stem_w_t,
            "drk_denoiser_bwd_bf16");
  CARGCHECK(dcs_slots == n_layers + 1 || (dcs_slots == 2 && !partials), "drk_denoiser_bwd_bf16 (dcs: n_layers + 1 slots, or 2 "
            "without partials)");
  CARGCHECK(!partials == !acc && !stem_dw == !stem_db && (!stem_dw || partials), "drk_denoiser_bwd_bf16 (partials with acc, stem_dw with stem_db; the stem pair only with partials)");
  const long M = (long)B * L, n = n_layers;
  const long W = drk_denoiser_bwd_bf16_slab_floats(n_layers);
  const char* wt = (const char*)wstream_t;
  char* d = (char*)dcs;
  auto dc = [&](long i) { return d + (dcs_slots == 2 ? (i & 1) : i) * M * 256; };
  auto slot = [&](long i) { return partials ? partials + i * CH_SLOT : nullptr; };
  TRY(drk_head_bwd_bf16(dlogits, y, head2_w, dc(n), slot(n + 1), W, B, L, stream));
  if (drk_denoiser_is_fused(L, n_layers, flags)) {
    // one sequence per workgroup, final_conv.0's input gradient and all residual layers in ONE launch; its column-sum
    // partials come as two slabs per sequence, drk_head_bwd_bf16's as one per 128 rows: two ordered reductions
    TRY(drk_train_bwd_fused_bf16(wstream_t, vecs, tbias, feats, feats + M * DRK_C, masks, means, rstds, dcs, dcs_slots, gr,
                                 partials, W, B, L, n_layers, dilations, stream));
    if (partials) {
      const long head_off = (n + 1) * CH_SLOT;
      const long nown = 2L * B > (M + 127) / 128 ? 2L * B : (M + 127) / 128;
      float* tmp = partials + nown * W;             // the workspace's tail (drk_denoiser_bwd_bf16_slabs)
      TRY(drk_slab_reduce2(partials, 2 * B, W, acc, head_off, tmp, stream));
      TRY(drk_slab_reduce(partials + head_off, (int)((M + 127) / 128), W, acc + head_off, W - head_off, stream));
      if (stem_db) TRY(drk_colsum128(gr, stem_db, 1, M, stream));
    }
    if (stem_dw || dx) TRY(drk_stem_bwd(gr, x, x_soft, stem_w_t, stem_dw, dx, B, L, stream));
    return DRK_OK;
  }
  TRY(drk_layer_bwd_bf16(dc(n), wt + 9 * n * CH_TAP_BYTES, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, gr,
                         masks + (n - 1) * M * DRK_C, dc(n - 1), slot(n), W, B, L, 1, 0, 0, stream));
  for (long i = n - 1; i >= 0; --i)
    TRY(drk_layer_bwd_bf16(dc(i), wt + 9 * i * CH_TAP_BYTES, feats + i * M * DRK_C, tbias + i * DRK_C, vecs + (3 * i + 1) * DRK_C,
                           means + i * M, rstds + i * M, gr, gr, i ? masks + (i - 1) * M * DRK_C : nullptr, i ? dc(i - 1) : nullptr,
                           slot(i), W, B, L, 9, dilations[i], 1, stream));
  TRY(drk_relu_bwd(gr, feats, gr, M * DRK_C, stream));
  if (partials) {
    TRY(drk_slab_reduce(partials, (int)((M + 127) / 128), W, acc, W, stream));
    if (stem_db) TRY(drk_colsum128(gr, stem_db, 1, M, stream));
  }
  if (stem_dw || dx) TRY(drk_stem_bwd(gr, x, x_soft, stem_w_t, stem_dw, dx, B, L, stream));
  return DRK_OK;
}
