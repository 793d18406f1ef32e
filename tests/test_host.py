"""Host-side logic and the C-ABI surface.  CPU only: nothing here launches a kernel."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import drakes_amd
from conftest import ROOT, load_golden
from drakes_amd import _lib, noise_schedule, ops
from drakes_amd.det_init import backbone_shapes, det_state_dict


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'drakes_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(drk_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert 'drk_reverse_step_hard' in names and 'drk_conv_fwd_f32' in names
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/drakes_hip.h but not exported'
    # and the binding table covers every compute entry point
    assert set(_lib.SIGNATURES) | set(_lib.HOST_QUERIES) == set(names) - {'drk_version', 'drk_last_error'}
    assert lib.drk_version() >= 100


def test_no_cpu_fallback():
    x = torch.zeros(2, 8, dtype=torch.int64)
    lg = torch.zeros(2, 8, 5)
    with pytest.raises(_lib.DrakesHipError):
        ops.reverse_step_hard(lg, x, 0.5, 0.4)
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=8, steps=2))
    with pytest.raises(_lib.DrakesHipError):
        m._sample(eval_sp_size=2)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'drakes_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f


def test_state_dict_keys_match_reference_layout():
    m = drakes_amd.Diffusion(drakes_amd.default_config())
    sd = m.state_dict()
    shapes = backbone_shapes()
    assert set(sd) == set(shapes) and len(sd) == 129
    for k, s in shapes.items():
        assert tuple(sd[k].shape) == s, k
    assert not m.backbone.time_embedder[0].W.requires_grad
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 3326597
    m.load_state_dict(det_state_dict(3), strict=True)
    # Lightning checkpoints carry the same keys under ['state_dict']
    m.load_state_dict({'state_dict': det_state_dict(4)}['state_dict'], strict=True)
    # deep-copied conv stacks start identical, like the reference (dnaconv.py:58)
    m2 = drakes_amd.Diffusion(drakes_amd.default_config())
    assert torch.equal(m2.backbone.convs[0].weight, m2.backbone.convs[3].weight)
    assert [m2.backbone.dilation(i) for i in (0, 7, 8, 11, 12, 16, 19)] == [1, 1, 4, 4, 16, 64, 64]


def test_constructor_errors_follow_reference():
    cfg = drakes_amd.default_config()
    cfg.backbone = 'dit'
    with pytest.raises(ValueError):
        drakes_amd.Diffusion(cfg)
    cfg = drakes_amd.default_config()
    cfg.parameterization = 'd3pm'
    with pytest.raises(AssertionError):
        drakes_amd.Diffusion(cfg)


def test_step_table_bits_match_reference():
    g = load_golden('f1_step_scalars.npz')
    table, sigma_last, timesteps, dt = noise_schedule.step_table(noise_schedule.LogLinearNoise(), 128)
    assert np.array_equal(table.numpy().view(np.uint32), g['table_bits'])
    assert np.array_equal(sigma_last.numpy().reshape(1).view(np.uint32), g['sigma_last_bits'])
    assert abs(dt - (1 - 1e-5) / 128) == 0


def test_weight_packing_is_a_permutation():
    w = torch.arange(128 * 128 * 9, dtype=torch.float32).reshape(128, 128, 9)
    pk = ops.pack_conv_weight_f32(w)
    assert pk.shape == (9, 32, 128, 4)
    assert pk[3, 5, 77, 2] == w[77, 4 * 5 + 2, 3]
    ws = torch.arange(128 * 5 * 9, dtype=torch.float32).reshape(128, 5, 9)
    assert ops.pack_stem_weight(ws)[8, 4, 100] == ws[100, 4, 8]


def test_fused_sampler_stem_fragments():
    """fastpath._frag_stem: lane 16g + n of co tile ct, row tile s, k-step ks holds W[32ct+16s+n][v=j][tap=4ks+g]."""
    from drakes_amd import fastpath
    ws = (torch.arange(128 * 5 * 9, dtype=torch.float32).reshape(128, 5, 9) % 251) - 125   # exact in bf16
    fr = fastpath._frag_stem(ws).float()
    assert fr.shape == (3, 4, 2, 64, 8)
    for ks, ct, s, g, n, j in [(0, 0, 0, 0, 0, 0), (1, 3, 1, 2, 9, 4), (2, 1, 0, 0, 15, 3), (0, 2, 1, 3, 7, 1)]:
        assert fr[ks, ct, s, 16 * g + n, j] == ws[32 * ct + 16 * s + n, j, 4 * ks + g]
    assert float(fr[:, :, :, :, 5:].abs().max()) == 0            # one-hot slots 5..7 do not exist
    assert float(fr[2, :, :, 16:, :].abs().max()) == 0           # taps 9..11 do not exist
    assert fr.abs().sum() == ws.abs().sum()                      # nothing lost, nothing duplicated


def test_sample_prior_and_sigma_processing():
    m = drakes_amd.Diffusion(drakes_amd.default_config())
    x = m._sample_prior(3, 200)
    assert x.dtype == torch.int64 and x.shape == (3, 200) and bool((x == 4).all())
    s = torch.tensor([[0.3], [0.7]])
    assert torch.equal(m._process_sigma(s), torch.zeros(2))
    m.time_conditioning = True
    assert torch.equal(m._process_sigma(s), torch.tensor([0.3, 0.7]))


def test_narrow_model_parameter_images():
    """hidden_dim < 128 (reference: models/dnaconv.py:46-63 takes it from args): the reference's parameter shapes, and
    a 128-lane image per parameter whose slices give the parameter back."""
    import drakes_amd
    m = drakes_amd.Diffusion(drakes_amd.default_config(hidden_dim=32))
    bb = m.backbone
    sd = m.state_dict()
    assert sd['backbone.linear.weight'].shape == (32, 5, 9) and sd['backbone.final_conv.2.weight'].shape == (5, 32, 1)
    assert sd['backbone.time_embedder.0.W'].shape == (16,) and sd['backbone.norms.19.weight'].shape == (32,)
    full = dict(drakes_amd.Diffusion(drakes_amd.default_config()).backbone.named_parameters())
    for name, p in bb.named_parameters():
        w = bb.wide(name, p.detach())
        assert w.shape == full[name].shape, name
        assert float(w.double().abs().sum()) == float(p.detach().double().abs().sum()), name   # nothing but zeros added
        if name != 'time_embedder.0.W':
            assert torch.equal(bb.narrow(name, w), p.detach()), name
    with pytest.raises(NotImplementedError):
        drakes_amd.Diffusion(drakes_amd.default_config(hidden_dim=20))


def test_native_denoiser_entry_points_reject_bad_arguments():
    """Argument errors of the round-3 entry points come back as DRK_EINVAL before anything is launched (so this runs
    without a GPU): the pointers below are never dereferenced."""
    lib = _lib.load()
    P = 0x10000                                   # any non-null, 16-byte aligned value
    dil = (ctypes.c_int * 20)(*([1] * 20))
    dp = ctypes.cast(dil, ctypes.c_void_p)

    def fwd(x=P, feats=P, B=2, L=200, n=20, dilations=dp, flags=0):
        return ('drk_denoiser_fwd_bf16', x, None if x is None else None, P, P, P, P, P, P, P, feats, P, P, P, P, P, P, B, L, n,
                dilations, flags, None)

    def bwd(slots=21, partials=P, acc=P, stem_dw=P, stem_db=P, B=2, L=200, n=20, flags=0, dlogits=P):
        return ('drk_denoiser_bwd_bf16', dlogits, P, P, P, P, P, P, P, P, P, P, slots, P, partials, acc, P, None, P, stem_dw,
                stem_db, P, B, L, n, dp, flags, None)

    bad = [
        (fwd(flags=2), 'unknown flag'),
        (fwd(feats=None), 'bad arguments'),
        (fwd(x=None), 'bad arguments'),             # neither x nor x_soft
        (fwd(n=0), 'bad arguments'),
        (fwd(dilations=None), 'bad arguments'),
        (fwd(L=0), 'bad shape'),
        (fwd(B=-1), 'bad shape'),
        (bwd(flags=4), 'unknown flag'),
        (bwd(dlogits=None), 'bad arguments'),
        (bwd(slots=3), 'dcs'),
        (bwd(slots=2), 'dcs'),                      # two slots only without parameter-gradient partials
        (bwd(acc=None), 'partials with acc'),
        (bwd(stem_db=None), 'stem_dw with stem_db'),
        (bwd(partials=None, acc=None, slots=2), 'only with partials'),
        (bwd(L=-3), 'bad shape'),
        (('drk_train_fwd_fused_bf16', P, P, P, P, P, P, P, P, P, P, 2, 225, 20, dp, None), 'L <= 224'),
        (('drk_train_fwd_fused_bf16', P, P, P, P, P, P, P, P, P, P, 2, 200, 21, dp, None), 'L <= 224'),
        (('drk_train_fwd_fused_bf16', P, P, P, P, None, P, P, P, P, P, 2, 200, 20, dp, None), 'null pointer'),
        (('drk_train_fwd_fused_bf16', P, P, P, P, P + 4, P, P, P, P, P, 2, 200, 20, dp, None), 'aligned'),
        (('drk_train_bwd_fused_bf16', P, P, P, P, P, P, P, P, P, 21, P, None, 0, 2, 512, 20, dp, None), 'L <= 224'),
        (('drk_train_bwd_fused_bf16', P, P, P, P, P, P, P, P, P, 5, P, None, 0, 2, 200, 20, dp, None), 'dcs_slots'),
        (('drk_train_bwd_fused_bf16', P, P, P, P, P, P, P, P, P, 21, P, P, 512, 2, 200, 20, dp, None), 'partials'),
        (('drk_layer_bwd_bf16', P, P, None, None, None, None, None, P, P, P, None, None, 0, 2, 200, 9, 1, 0, None), 'come together'),
        (('drk_layer_bwd_bf16', P, P, None, None, None, None, None, P, P, P, P, None, 0, 2, 200, 9, 1, 0, None), 'distinct'),
        (('drk_layer_bwd_bf16', P, P, None, None, None, None, None, P, P, None, None, None, 0, 2, 200, 9, 1, 1, None), 'ln = 1'),
        (('drk_layer_bwd_bf16', P, P, None, None, None, None, None, P, P, None, None, None, 0, 2, 200, 4, 1, 0, None), 'bad arguments'),
        (('drk_layer_bwd_bf16', P, P, None, None, None, None, None, P, P, None, None, P, 256, 2, 200, 9, 1, 0, None), 'partials'),
        (('drk_head_bwd_bf16', P, P, P, P, P, 512, 2, 200, None), '776'),
        (('drk_head_bwd_bf16', P, P, None, P, None, 0, 2, 200, None), 'bad arguments'),
        (('drk_conv_wgrad_bf16_multi', P, P, -1, P, 2, 200, 9, 1, P, 1 << 40, None), 'bad arguments'),
        (('drk_conv_wgrad_bf16_multi', P, P, 3, P, 2, 200, 11, 1, P, 1 << 40, None), 'bad arguments'),
        (('drk_conv_wgrad_bf16_multi', P, P, 3, P, 2, 200, 9, 1, P, 16, None), 'workspace'),
        (('drk_conv_wgrad_bf16_multi', P, P, 3, P, 2, 200, 9, 1, P + 8, 1 << 40, None), 'workspace'),
        (('drk_slab_reduce', P, 4, 512, P, 510, None), 'n % 4'),
        (('drk_slab_reduce', P, 4, 510, P, 512, None), 'stride % 4'),
        (('drk_slab_reduce2', P, 40, 512, P, 512, None, None), 'tmp given'),
        (('drk_slab_reduce2', P + 4, 40, 512, P, 512, P, None), 'aligned'),
    ]
    for call, needle in bad:
        with pytest.raises(_lib.DrakesHipError, match=re.escape(needle)):
            _lib.call(*call)
    # an empty batch is a no-op of the whole sequence, not an error
    _lib.call(*fwd(B=0))
    _lib.call(*bwd(B=0))
    _lib.call('drk_conv_wgrad_bf16_multi', P, P, 0, P, 2, 200, 9, 1, P, 0, None)       # no recorded pairs: nothing to add
    # size queries: 0 for shapes that hold nothing; the fused layouts only within the fused kernels' limits
    assert lib.drk_denoiser_feats_floats(0, 200, 20, 0) == 0 and lib.drk_denoiser_masks_bytes(4, 200, 0, 0) == 0
    assert lib.drk_denoiser_is_fused(200, 20, 0) == 1 and lib.drk_denoiser_is_fused(225, 20, 0) == 0
    assert lib.drk_denoiser_is_fused(200, 21, 0) == 0 and lib.drk_denoiser_is_fused(200, 20, ops.denoiser_flags(True)) == 0
    assert lib.drk_denoiser_feats_floats(4, 512, 20, 0) == 21 * 4 * 512 * 128
    assert lib.drk_denoiser_feats_floats(4, 200, 20, 0) == 4 * 200 * 128 + lib.drk_train_fused_featn_floats(4, 20)
    assert lib.drk_denoiser_masks_bytes(4, 200, 20, ops.denoiser_flags(True)) == 20 * 4 * 200 * 128
    assert lib.drk_denoiser_bwd_bf16_slab_floats(20) == 21 * 512 + lib.drk_head_bwd_bf16_slab_floats()
    assert lib.drk_denoiser_bwd_bf16_slabs(128, 200) == 256 + 16 + 1 and lib.drk_denoiser_bwd_bf16_slabs(2, 4096) == 64 + 4 + 1


def test_group_plan_covers_the_stack_once():
    """fastpath.group_plan (host logic of the grouped long-sequence forward): every layer in exactly one group, halo groups
    only over dilations 1..4, stride groups closed under their stride, nothing planned where an image would overflow."""
    from drakes_amd import fastpath

    class Stack:
        def __init__(self, dil):
            self.dil, self.num_layers = dil, len(dil)

        def dilation(self, i):
            return self.dil[i]

    std = Stack([1] * 8 + [4] * 4 + [16] * 4 + [64] * 4)
    assert fastpath.group_plan(std, 1024) == [(0, 4, 1, 18), (4, 4, 1, 18), (8, 12, 4, 16)]
    assert fastpath.group_plan(std, 2048) is None and fastpath.group_plan(std, 1025) is None
    lib = _lib.load()
    for L in (225, 300, 640, 777, 1000, 1024):
        for hl, ht in [(None, None), (1, 14), (2, 16), (8, 18), (12, 16)]:
            plan = fastpath.group_plan(std, L, hl, ht)
            assert [l for g in plan for l in range(g[0], g[0] + g[1])] == list(range(20)), (L, plan)
            for l0, n, s, t16 in plan:
                d = std.dil[l0:l0 + n]
                if s == 1:
                    assert all(1 <= x <= 4 for x in d) and lib.drk_layer_group_bf16_rows(n, t16) == 16 * t16 - 8 * sum(d) >= 16
                else:
                    assert t16 == 16 and n <= 12 and all(x % s == 0 and x // s in (1, 2, 3, 4, 16) for x in d) and -(-L // s) <= 256
    cyc = Stack([1, 1, 4, 16, 64] * 2)                      # a stack that cycles its dilations: groups stay short
    assert fastpath.group_plan(cyc, 1024) == [(0, 2, 1, 18), (2, 3, 4, 16), (5, 2, 1, 18), (7, 3, 4, 16)]
    odd = Stack([1, 5, 25])
    assert fastpath.group_plan(odd, 1024) == [(0, 1, 1, 18), (1, 1, 5, 16), (2, 1, 25, 16)]
    assert lib.drk_layer_group_bf16_rows(4, 18) == 256 and lib.drk_layer_group_bf16_rows(2, 14) == 208
    assert lib.drk_layer_group_bf16_rows(13, 16) == 0 and lib.drk_layer_group_bf16_rows(2, 15) == 0


def test_group_entry_point_argument_errors():
    """drk_layer_group_bf16 rejects bad arguments before anything is launched (runs without a GPU)."""
    P = 0x10000
    d1 = (ctypes.c_int * 4)(1, 1, 1, 1)
    d4 = (ctypes.c_int * 4)(4, 4, 16, 64)

    def call(feat_in=P, x_tok=None, stem_w=None, stem_b=None, feat_out=P + 0x100000, h_out=None, logits=None, head_w=None,
             head_b=None, B=2, L=1024, first=0, n=4, stride=1, t16=16, dil=d1):
        return _lib.call('drk_layer_group_bf16', feat_in, x_tok, stem_w, stem_b, feat_out, h_out, logits, head_w, head_b, P, P, P,
                         B, L, first, n, stride, t16, ctypes.cast(dil, ctypes.c_void_p), None)

    for kw, needle in [(dict(feat_in=None), 'one of feat_in / x_tok'), (dict(x_tok=P), 'one of feat_in / x_tok'),
                       (dict(feat_out=None), 'at least one output'), (dict(feat_in=None, x_tok=P), 'tokens come with'),
                       (dict(feat_in=None, x_tok=P, stem_w=P, stem_b=P, first=4), 'tokens come with'),
                       (dict(feat_in=None, x_tok=P, stem_w=P, stem_b=P, stride=4, dil=d4), 'tokens come with'),
                       (dict(logits=P), 'logits come with'), (dict(feat_in=P + 4), 'aligned'), (dict(n=0), '1..12 layers'),
                       (dict(n=13), '1..12 layers'), (dict(t16=15), 'tiles16'), (dict(stride=0), 'stride >= 1'),
                       (dict(stride=4, dil=d1), 'multiple of stride'), (dict(dil=d4), 'dilation / stride'),
                       (dict(stride=4, dil=d4, t16=18), 'tiles16 = 16'), (dict(stride=4, dil=d4, L=1025), '<= 256'),
                       (dict(feat_out=P), 'feat_out != feat_in'), (dict(L=0), 'bad shape'), (dict(B=-1), 'bad shape')]:
        with pytest.raises(_lib.DrakesHipError, match=re.escape(needle)):
            call(**kw)
    call(B=0)                                             # an empty batch is a no-op
