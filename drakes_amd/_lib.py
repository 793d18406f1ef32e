"""ctypes binding of libdrakes_hip.so (the C ABI declared in include/drakes_hip.h).

There is no fallback: if the library is missing or a call fails, this raises.  Build it with
`python -m drakes_amd.build` (hipcc cross-compiles gfx950 without a GPU).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRK_LIB: experiment builds (tools/exp.py); the product always loads the in-tree library
LIB_PATH = os.environ.get('DRK_LIB') or os.path.join(_HERE, 'libdrakes_hip.so')

_P = c_void_p

# name -> argtypes; every function returns int (0 = ok).  Kept in one table so that the CPU test
# suite can check that the library exports exactly what include/drakes_hip.h declares.
SIGNATURES = {
    'drk_step_uniforms': [_P, c_int, c_int, c_uint64, c_uint32, c_uint64, _P],
    'drk_subs_fwd': [_P, _P, _P, _P, c_int, c_int, _P],
    'drk_reverse_step_hard': [_P, _P, _P, _P, c_int, c_int, c_float, c_float, _P, c_uint64, c_uint32, c_uint64, _P],
    'drk_reverse_step_hard_table': [_P, _P, _P, c_int, c_int, _P, _P, _P, _P],
    'drk_reverse_step_hard_wide': [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_uint64,
                                   c_uint32, c_uint64, _P],
    'drk_reverse_step_soft_wide_fwd': [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float,
                                       c_uint64, c_uint32, c_uint64, _P],
    'drk_reverse_step_soft_wide_bwd': [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float,
                                       c_float, c_uint64, c_uint32, c_uint64, _P],
    'drk_denoise_argmax': [_P, _P, _P, c_int, c_int, _P],
    'drk_q_xt': [_P, _P, _P, c_int, c_int, c_float, _P, c_uint64, c_uint32, c_uint64, _P],
    'drk_likelihood_accum': [_P, _P, _P, c_float, _P, c_int, c_int, _P],
    'drk_time_embed_fwd': [_P, c_int, _P, _P, _P, _P, _P],
    'drk_time_bias_fwd': [_P, c_int, _P, _P, c_int, _P, _P],
    'drk_stem_fwd': [_P, _P, _P, _P, _P, c_int, c_int, _P],
    'drk_ln_fwd': [_P, _P, c_long, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    'drk_conv_fwd_f32': [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    'drk_conv_fwd_x3': [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    'drk_mask_mul': [_P, _P, _P, c_long, _P],
    'drk_relu_bwd': [_P, _P, _P, c_long, _P],
    'drk_colsum128': [_P, _P, c_long, c_long, _P],
    'drk_ln_bwd': [_P, _P, _P, c_long, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    'drk_conv_wgrad_f32': [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_long, _P],
    'drk_conv_wgrad_f32_static': [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    'drk_head_out_bwd': [_P, _P, _P, _P, _P, _P, c_int, c_int, _P],
    'drk_stem_bwd': [_P, _P, _P, _P, _P, _P, c_int, c_int, _P],
    'drk_time_bwd': [_P, _P, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P],
    'drk_subs_bwd': [_P, _P, _P, _P, _P, c_int, c_int, _P],
    'drk_reverse_step_soft_fwd': [_P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, _P, c_float, c_float, c_uint64,
                                  c_uint32, c_uint64, _P],
    'drk_reverse_step_soft_bwd': [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, _P, c_float, c_float,
                                  c_uint64, c_uint32, c_uint64, _P],
    'drk_kl_step_fwd': [_P, _P, _P, _P, _P, c_float, _P, c_int, c_int, _P],
    'drk_kl_step_bwd': [_P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, _P],
    'drk_head_out_fwd': [_P, _P, _P, _P, c_int, c_int, _P],
    'drk_layer_bf16': [_P, _P, _P, _P, _P, _P, _P, c_long, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    'drk_ln_fwd_bf16': [_P, _P, c_long, _P, _P, _P, _P, _P, c_int, c_int, _P],
    'drk_head_bf16': [_P, _P, _P, _P, _P, _P, c_int, c_int, _P],
    'drk_stem_ln_bf16': [_P, _P, _P, _P, c_long, _P, _P, _P, _P, _P, _P, c_int, c_int, _P],
    'drk_gate_bf16': [_P, _P, _P, _P, _P, c_int, c_int, _P, c_long, _P],
    'drk_conv_wgrad_bf16': [_P, _P, _P, c_int, c_int, c_int, c_int, _P, c_long, _P],
    'drk_layer_bwd_bf16': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_long, c_int, c_int, c_int, c_int, c_int, _P],
    'drk_head_bwd_bf16': [_P, _P, _P, _P, _P, c_long, c_int, c_int, _P],
    'drk_conv_wgrad_bf16_multi': [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_long, _P],
    'drk_stem_wgrad_multi': [_P, _P, _P, c_int, _P, _P, c_int, c_int, _P],
    'drk_slab_reduce': [_P, c_int, c_long, _P, c_long, _P],
    'drk_slab_reduce2': [_P, c_int, c_long, _P, c_long, _P, _P],
    'drk_denoiser_fwd_bf16': [_P] * 16 + [c_int, c_int, c_int, _P, c_int, _P],
    'drk_train_fwd_fused_bf16': [_P] * 10 + [c_int, c_int, c_int, _P, _P],
    'drk_denoiser_bwd_bf16': [_P] * 11 + [c_int] + [_P] * 9 + [c_int, c_int, c_int, _P, c_int, _P],
    'drk_train_bwd_fused_bf16': [_P] * 9 + [c_int, _P, _P, c_long, c_int, c_int, c_int, _P, _P],
    'drk_layer_group_bf16': [_P] * 12 + [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P],
    'drk_sample_bf16': [_P, _P, _P, _P, _P, c_long, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int,
                        c_uint64, c_uint64, c_int, _P],
}

HOST_QUERIES = {
    'drk_conv_wgrad_workspace_bytes': [c_int, c_int, c_int],
    'drk_conv_wgrad_bf16_workspace_bytes': [c_int, c_int, c_int],
    'drk_gate_bf16_workspace_bytes': [],
    'drk_conv_wgrad_bf16_multi_workspace_bytes': [c_int, c_int, c_int, c_int],
    'drk_head_bwd_bf16_slab_floats': [],
    'drk_denoiser_bwd_bf16_slab_floats': [c_int],
    'drk_denoiser_bwd_bf16_slabs': [c_int, c_int],
    'drk_denoiser_is_fused': [c_int, c_int, c_int],
    'drk_slab_reduce2_tmp_floats': [c_int, c_long],
    'drk_denoiser_feats_floats': [c_int, c_int, c_int, c_int],
    'drk_denoiser_masks_bytes': [c_int, c_int, c_int, c_int],
    'drk_train_fused_featn_floats': [c_int, c_int],
    'drk_train_fused_maskp_bytes': [c_int, c_int],
    'drk_layer_group_bf16_rows': [c_int, c_int],
}

_lib = None


class DrakesHipError(RuntimeError):
    pass


def load():
    """Loads the library once; raises DrakesHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DrakesHipError(
            f'{LIB_PATH} is missing: run `python -m drakes_amd.build`. '
            'drakes_amd has no CPU or eager-PyTorch fallback for the hot path.')
    lib = ctypes.CDLL(LIB_PATH)
    lib.drk_version.restype = c_int
    lib.drk_version.argtypes = []
    lib.drk_last_error.restype = c_char_p
    lib.drk_last_error.argtypes = []
    for name, argtypes in HOST_QUERIES.items():       # host-side size queries: return a byte count, launch nothing
        fn = getattr(lib, name)
        fn.restype = c_long
        fn.argtypes = argtypes
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise DrakesHipError(f'{name} failed ({rc}): {lib.drk_last_error().decode()}')
