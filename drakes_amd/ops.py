"""Tensor-level wrappers over the C ABI (include/drakes_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every op below is one call
into libdrakes_hip.so on raw device pointers.  Tensors must live on a ROCm device; nothing
falls back to torch kernels.
"""
import ctypes

import torch

from . import _lib

C = 128      # hidden_dim the kernels are written for
V = 5        # vocabulary incl. mask
MASK = 4


class _Profile:
    """HIP-event timing of the dominant kernel's launches, on the stream they are launched on
    (torch's current stream).  Off by default; bench.py switches it on over its timed region."""

    def __init__(self):
        self.on = False
        self._pairs = []
        self._n = 0
        self.every = 1
        self.name = None

    def enable(self, every=1):
        self.on, self.every, self._pairs, self._n, self.name = True, max(1, every), [], 0, None

    def bracket(self, name, flop, launch):
        """Runs launch(); when profiling, every `every`-th call is bracketed by an event pair."""
        if not self.on:
            return launch()
        self._n += 1
        if self._n % self.every:
            return launch()
        self.name = name
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = launch()
        b.record()
        self._pairs.append((a, b, flop))
        return r

    def disable(self):
        """-> (kernel name, launches timed, average ms per launch, flop per launch)."""
        self.on = False
        torch.cuda.synchronize()
        n = len(self._pairs)
        if n == 0:
            return (None, 0, None, 0)
        ms = sum(a.elapsed_time(b) for a, b, _ in self._pairs)
        fl = sum(f for _, _, f in self._pairs)
        self._pairs = []
        return (self.name, n, ms / n, fl / n)


PROFILE = _Profile()


def _dev_ptr(t, dtype, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.DrakesHipError(f'{name}: expected a tensor on the GPU (drakes_amd has no CPU path)')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: tensor must be contiguous')
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f(t, name):
    return _dev_ptr(t, torch.float32, name)


def _i(t, name):
    return _dev_ptr(t, torch.int64, name)


def step_uniforms(B, L, seed, step, sample_offset=0, device=None, out=None):
    u = out if out is not None else torch.empty(B, L, V, dtype=torch.float32, device=device)
    _lib.call('drk_step_uniforms', _f(u, 'u'), B, L, seed, step, sample_offset, _stream())
    return u


def subs_fwd(logits, x, out=None):
    """x: int64 [B,L] tokens or float32 [B,L,5] soft state."""
    B, L, _ = logits.shape
    out = out if out is not None else torch.empty_like(logits)
    if x.dtype == torch.int64:
        _lib.call('drk_subs_fwd', _f(logits, 'logits'), _i(x, 'x'), None, _f(out, 'logp'), B, L, _stream())
    else:
        _lib.call('drk_subs_fwd', _f(logits, 'logits'), None, _f(x, 'x_soft'), _f(out, 'logp'), B, L, _stream())
    return out


def reverse_step_hard(logits, x, mc_t, mc_s, seed=0, step=0, sample_offset=0, uniforms=None, out=None):
    B, L = x.shape
    out = out if out is not None else torch.empty_like(x)
    _lib.call('drk_reverse_step_hard', _f(logits, 'logits'), _i(x, 'x'), _f(uniforms, 'uniforms'), _i(out, 'x_next'),
              B, L, float(mc_t), float(mc_s), seed, step, sample_offset, _stream())
    return out


def denoise_argmax(logits, x, out=None):
    B, L = x.shape
    out = out if out is not None else torch.empty_like(x)
    _lib.call('drk_denoise_argmax', _f(logits, 'logits'), _i(x, 'x'), _i(out, 'x_out'), B, L, _stream())
    return out


def time_embed_fwd(sigma, W, lin_w, lin_b):
    nb = sigma.shape[0]
    temb = torch.empty(nb, C, dtype=torch.float32, device=sigma.device)
    _lib.call('drk_time_embed_fwd', _f(sigma, 'sigma'), nb, _f(W, 'W'), _f(lin_w, 'lin_w'), _f(lin_b, 'lin_b'),
              _f(temb, 'temb'), _stream())
    return temb


def time_bias_fwd(temb, dense_w, dense_b):
    nb, n_layers = temb.shape[0], dense_w.shape[0]
    tbias = torch.empty(nb, n_layers, C, dtype=torch.float32, device=temb.device)
    _lib.call('drk_time_bias_fwd', _f(temb, 'temb'), nb, _f(dense_w, 'dense_w'), _f(dense_b, 'dense_b'), n_layers,
              _f(tbias, 'tbias'), _stream())
    return tbias


def stem_fwd(x, w_t, bias, out=None):
    B, L = x.shape[0], x.shape[1]
    out = out if out is not None else torch.empty(B, L, C, dtype=torch.float32, device=x.device)
    if x.dtype == torch.int64:
        _lib.call('drk_stem_fwd', _i(x, 'x'), None, _f(w_t, 'w_t'), _f(bias, 'bias'), _f(out, 'feat'), B, L, _stream())
    else:
        _lib.call('drk_stem_fwd', None, _f(x, 'x_soft'), _f(w_t, 'w_t'), _f(bias, 'bias'), _f(out, 'feat'), B, L,
                  _stream())
    return out


def ln_fwd(feat, tbias_layer, tb_stride, gamma, beta, out=None, mean=None, rstd=None):
    """tbias_layer: a view starting at this layer's 128 floats of batch row 0; tb_stride: floats
    between batch rows (0 = shared)."""
    B, L, _ = feat.shape
    out = out if out is not None else torch.empty_like(feat)
    tb = ctypes.c_void_p(tbias_layer.data_ptr()) if tbias_layer is not None else None
    _lib.call('drk_ln_fwd', _f(feat, 'feat'), tb, tb_stride, _f(gamma, 'gamma'), _f(beta, 'beta'), _f(out, 'h'),
              _f(mean, 'mean'), _f(rstd, 'rstd'), B, L, _stream())
    return out


def conv_fwd_f32(h, w_pk, bias, resid, dilation, relu=True, out=None):
    B, L, _ = h.shape
    taps = w_pk.shape[0]
    out = out if out is not None else torch.empty_like(h)
    args = (_f(h, 'h'), _f(w_pk, 'w_pk'), _f(bias, 'bias'), _f(resid, 'resid'), _f(out, 'out'), B, L, taps, dilation,
            1 if relu else 0, _stream())
    if taps == 9:
        PROFILE.bracket('k_conv_f32<2>', 2.0 * B * L * C * C * taps, lambda: _lib.call('drk_conv_fwd_f32', *args))
    else:
        _lib.call('drk_conv_fwd_f32', *args)
    return out


def head_out_fwd(y, w, bias, out=None):
    B, L, _ = y.shape
    out = out if out is not None else torch.empty(B, L, V, dtype=torch.float32, device=y.device)
    _lib.call('drk_head_out_fwd', _f(y, 'y'), _f(w, 'w'), _f(bias, 'bias'), _f(out, 'logits'), B, L, _stream())
    return out


# ---- weight layouts the kernels read (pure index permutations) --------------------------------
def pack_conv_weight_f32(w):
    """[co][ci][taps] -> [taps][ci/4][co][4] (drk_conv_fwd_f32's w_pk)."""
    co, ci, taps = w.shape
    return w.detach().permute(2, 1, 0).reshape(taps, ci // 4, 4, co).permute(0, 1, 3, 2).contiguous()


def pack_stem_weight(w):
    """[co][5][9] -> [9][5][co]."""
    return w.detach().permute(2, 1, 0).contiguous()
