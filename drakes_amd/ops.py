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


def _mc_rows(mc_rows, B, width):
    """Optional per-row move chances: float32 [B, width] on the device (width 2 = (mc_t, mc_s); 1 for q_xt)."""
    if mc_rows is None:
        return None
    if mc_rows.shape != (B, width):
        raise ValueError(f'mc_rows: expected shape {(B, width)}, got {tuple(mc_rows.shape)}')
    return _f(mc_rows, 'mc_rows')


def reverse_step_hard(logits, x, mc_t, mc_s, seed=0, step=0, sample_offset=0, uniforms=None, out=None, mc_rows=None):
    """mc_rows: optional float32 [B,2] = (mc_t, mc_s) per row (per-row timesteps); then mc_t / mc_s are ignored."""
    B, L = x.shape
    out = out if out is not None else torch.empty_like(x)
    _lib.call('drk_reverse_step_hard', _f(logits, 'logits'), _i(x, 'x'), _f(uniforms, 'uniforms'), _i(out, 'x_next'),
              B, L, float(mc_t), float(mc_s), _mc_rows(mc_rows, B, 2), seed, step, sample_offset, _stream())
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


def ln_fwd(feat, tbias_layer, tb_stride, gamma, beta, out=None, mean=None, rstd=None, width=C):
    """tbias_layer: a view starting at this layer's 128 floats of batch row 0; tb_stride: floats
    between batch rows (0 = shared).  width: the model's hidden_dim when it is narrower than the 128 lanes."""
    B, L, _ = feat.shape
    out = out if out is not None else torch.empty_like(feat)
    tb = ctypes.c_void_p(tbias_layer.data_ptr()) if tbias_layer is not None else None
    _lib.call('drk_ln_fwd', _f(feat, 'feat'), tb, tb_stride, _f(gamma, 'gamma'), _f(beta, 'beta'), _f(out, 'h'),
              _f(mean, 'mean'), _f(rstd, 'rstd'), B, L, width, _stream())
    return out


def conv_fwd_f32(h, w_pk, bias, resid, dilation, relu=True, out=None, relu_mask=None):
    B, L, _ = h.shape
    taps = w_pk.shape[0]
    out = out if out is not None else torch.empty_like(h)
    args = (_f(h, 'h'), _f(w_pk, 'w_pk'), _f(bias, 'bias'), _f(resid, 'resid'), _f(out, 'out'),
            _dev_ptr(relu_mask, torch.uint8, 'relu_mask'), B, L, taps, dilation, 1 if relu else 0, _stream())
    if taps == 9:
        PROFILE.bracket('k_conv_f32_lds' if B * L >= 16384 else 'k_conv_f32<1>', 2.0 * B * L * C * C * taps, lambda: _lib.call('drk_conv_fwd_f32', *args))
    else:
        _lib.call('drk_conv_fwd_f32', *args)
    return out


def split3_bf16(w):
    """fp32 -> three bf16 planes whose sum is w to 2^-24 relative (drk_conv_x3.hip splits activations the same way)."""
    w = w.detach().float()
    p0 = w.to(torch.bfloat16)
    r1 = w - p0.float()
    p1 = r1.to(torch.bfloat16)
    p2 = (r1 - p1.float()).to(torch.bfloat16)
    return p0, p1, p2


def pack_conv_weight_x3(w):
    """[co][ci][taps] fp32 -> bf16 [taps][3 planes][4][4][2][64][8] (drk_conv_fwd_x3's w3)."""
    from .fastpath import _frag_tap
    planes = split3_bf16(w)
    return torch.stack([torch.stack([_frag_tap(p[:, :, t].float()) for p in planes]) for t in range(w.shape[2])]).contiguous()


def conv_fwd_x3(h, w3, bias, resid, dilation, relu=True, out=None, relu_mask=None):
    """drk_conv_fwd_f32's operation at fp32 grade on the bf16 matrix cores (three bf16 planes per operand)."""
    B, L, _ = h.shape
    taps = w3.shape[0]
    out = out if out is not None else torch.empty_like(h)
    args = (_f(h, 'h'), _dev_ptr(w3, torch.bfloat16, 'w3'), _f(bias, 'bias'), _f(resid, 'resid'), _f(out, 'out'),
            _dev_ptr(relu_mask, torch.uint8, 'relu_mask'), B, L, taps, dilation, 1 if relu else 0, _stream())
    if taps == 9:
        PROFILE.bracket('k_conv_x3', 2.0 * B * L * C * C * taps, lambda: _lib.call('drk_conv_fwd_x3', *args))
    else:
        _lib.call('drk_conv_fwd_x3', *args)
    return out


def head_out_fwd(y, w, bias, out=None):
    B, L, _ = y.shape
    out = out if out is not None else torch.empty(B, L, V, dtype=torch.float32, device=y.device)
    _lib.call('drk_head_out_fwd', _f(y, 'y'), _f(w, 'w'), _f(bias, 'bias'), _f(out, 'logits'), B, L, _stream())
    return out


def conv_wgrad_f32_static(h, dc, dwt, dilation, dbias=None):
    """conv_wgrad_f32 through the allocation-free entry point (the partial slabs live inside the library)."""
    B, L, _ = h.shape
    _lib.call('drk_conv_wgrad_f32_static', _f(h, 'h'), _f(dc, 'dc'), _f(dwt, 'dwt'), _f(dbias, 'dbias'), B, L, dwt.shape[0],
              dilation, _stream())


# ---- backward ops ---------------------------------------------------------------------------------
def mask_mul(g, mask, out=None):
    out = out if out is not None else torch.empty_like(g)
    _lib.call('drk_mask_mul', _f(g, 'g'), _dev_ptr(mask, torch.uint8, 'mask'), _f(out, 'out'), g.numel(), _stream())
    return out


def relu_bwd(g, y, out=None):
    out = out if out is not None else torch.empty_like(g)
    _lib.call('drk_relu_bwd', _f(g, 'g'), _f(y, 'y'), _f(out, 'out'), g.numel(), _stream())
    return out


def colsum128(x, n_seg, seg_rows, out):
    """out [n_seg, 128] += column sums of x viewed as [n_seg * seg_rows, 128]."""
    _lib.call('drk_colsum128', _f(x, 'x'), _f(out, 'out'), n_seg, seg_rows, _stream())
    return out


def ln_bwd(dh, feat, tbias_layer, tb_stride, gamma, mean, rstd, g_resid, dxin, dfeat, dgamma, dbeta, dtb_shared=None,
           next_mask=None, dc_next=None, dbias_next=None, width=C):
    """dxin may be None (not stored).  next_mask / dc_next / dbias_next: the fused hand-over to the layer below (bf16
    path): dc_next = bf16(dfeat * next_mask), dbias_next += its column sums."""
    B, L, _ = feat.shape
    tb = ctypes.c_void_p(tbias_layer.data_ptr()) if tbias_layer is not None else None
    _lib.call('drk_ln_bwd', _f(dh, 'dh'), _f(feat, 'feat'), tb, tb_stride, _f(gamma, 'gamma'), _f(mean, 'mean'),
              _f(rstd, 'rstd'), _f(g_resid, 'g_resid'), _f(dxin, 'dxin'), _f(dfeat, 'dfeat'), _f(dgamma, 'dgamma'),
              _f(dbeta, 'dbeta'), _f(dtb_shared, 'dtbias_shared'), _dev_ptr(next_mask, torch.uint8, 'next_mask'),
              _dev_ptr(dc_next, torch.bfloat16, 'dc_next'), _f(dbias_next, 'dbias_next'), B, L, width, _stream())


_WORKSPACE = {}


def workspace(nbytes, device):
    """A scratch tensor of at least `nbytes` (split-K partial slabs) per (device, current stream).  Grown on demand and
    reused by every call on that stream, which orders them; work on another stream (a second model's backward, side-stream
    weight gradients) gets its own buffer instead of racing on this one."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _WORKSPACE.get(key)
    if t is None or t.numel() * 4 < nbytes:
        t = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WORKSPACE[key] = t
    return t


def conv_wgrad_f32(h, dc, dwt, dilation, dbias=None, ws=None):
    """dwt [taps,128(ci),128(co)] += ...; dbias [128] += column sums of dc (optional).  Deterministic: per-chunk partial
    slabs in the workspace `ws`, summed in a fixed order."""
    B, L, _ = h.shape
    taps = dwt.shape[0]
    need = _lib.load().drk_conv_wgrad_workspace_bytes(B, L, taps)
    if ws is None:
        ws = workspace(need, h.device)
    _lib.call('drk_conv_wgrad_f32', _f(h, 'h'), _f(dc, 'dc'), _f(dwt, 'dwt'), _f(dbias, 'dbias'), B, L, taps, dilation,
              _f(ws, 'workspace'), ws.numel() * 4, _stream())


def head_out_bwd(dlogits, y, w, dy, dw, db):
    B, L, _ = y.shape
    _lib.call('drk_head_out_bwd', _f(dlogits, 'dlogits'), _f(y, 'y'), _f(w, 'w'), _f(dy, 'dy'), _f(dw, 'dw'),
              _f(db, 'db'), B, L, _stream())


def stem_bwd(dpre, x, w_t, dw_t, dx):
    B, L = x.shape[0], x.shape[1]
    if x.dtype == torch.int64:
        _lib.call('drk_stem_bwd', _f(dpre, 'dpre'), _i(x, 'x'), None, _f(w_t, 'w_t'), _f(dw_t, 'dw_t'), None, B, L,
                  _stream())
    else:
        _lib.call('drk_stem_bwd', _f(dpre, 'dpre'), None, _f(x, 'x_soft'), _f(w_t, 'w_t'), _f(dw_t, 'dw_t'),
                  _f(dx, 'dx'), B, L, _stream())


def time_bwd(dtbias, sigma, W, temb, dense_w, d_dense_w, d_dense_b, d_lin_w, d_lin_b, dtemb):
    """dtemb: zeroed float32 [nb,128] scratch."""
    nb, n_layers = dtbias.shape[0], dtbias.shape[1]
    _lib.call('drk_time_bwd', _f(dtbias, 'dtbias'), _f(sigma, 'sigma'), nb, _f(W, 'W'), _f(temb, 'temb'),
              _f(dense_w, 'dense_w'), n_layers, _f(d_dense_w, 'd_dense_w'), _f(d_dense_b, 'd_dense_b'),
              _f(d_lin_w, 'd_lin_w'), _f(d_lin_b, 'd_lin_b'), _f(dtemb, 'dtemb'), _stream())


def subs_bwd(logits, x, dlogp, out=None):
    B, L, _ = logits.shape
    out = out if out is not None else torch.empty_like(logits)
    if x.dtype == torch.int64:
        _lib.call('drk_subs_bwd', _f(logits, 'logits'), _i(x, 'x'), None, _f(dlogp, 'dlogp'), _f(out, 'dlogits'), B, L,
                  _stream())
    else:
        _lib.call('drk_subs_bwd', _f(logits, 'logits'), None, _f(x, 'x_soft'), _f(dlogp, 'dlogp'), _f(out, 'dlogits'),
                  B, L, _stream())
    return out


def reverse_step_soft_fwd(logits, x_soft, mc_t, mc_s, temp, copy_flag_temp, seed=0, step=0, sample_offset=0,
                          uniforms=None, mc_rows=None):
    B, L, _ = x_soft.shape
    x_next = torch.empty_like(x_soft)
    flag = torch.empty(B, L, dtype=torch.float32, device=x_soft.device)
    _lib.call('drk_reverse_step_soft_fwd', _f(logits, 'logits'), _f(x_soft, 'x_soft'), _f(uniforms, 'uniforms'),
              _f(x_next, 'x_next'), _f(flag, 'flag'), B, L, float(mc_t), float(mc_s), _mc_rows(mc_rows, B, 2), float(temp),
              float(copy_flag_temp) if copy_flag_temp is not None else 0.0, seed, step, sample_offset, _stream())
    return x_next, flag


def reverse_step_soft_bwd(logits, x_soft, grad_x_next, grad_flag, mc_t, mc_s, temp, copy_flag_temp, seed=0, step=0,
                          sample_offset=0, uniforms=None, mc_rows=None):
    B, L, _ = x_soft.shape
    dlogits = torch.empty_like(logits)
    dx = torch.empty_like(x_soft)
    _lib.call('drk_reverse_step_soft_bwd', _f(logits, 'logits'), _f(x_soft, 'x_soft'), _f(uniforms, 'uniforms'),
              _f(grad_x_next, 'grad_x_next'), _f(grad_flag, 'grad_flag'), _f(dlogits, 'dlogits'), _f(dx, 'dx'), B, L,
              float(mc_t), float(mc_s), _mc_rows(mc_rows, B, 2), float(temp),
              float(copy_flag_temp) if copy_flag_temp is not None else 0.0, seed, step, sample_offset, _stream())
    return dlogits, dx


# ---- bf16 layer / backward ops (the differentiable bf16 path) ------------------------------------------
def _b16(t, name):
    return _dev_ptr(t, torch.bfloat16, name)


def _raw(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def ln_fwd_bf16(feat, tbias_layer, tb_stride, gamma, beta, out=None, mean=None, rstd=None):
    """h = bf16(LayerNorm(feat + tbias) * gamma + beta); optional mean / rstd [B*L] for the backward pass."""
    B, L, _ = feat.shape
    out = out if out is not None else torch.empty(B, L, C, dtype=torch.bfloat16, device=feat.device)
    _lib.call('drk_ln_fwd_bf16', _f(feat, 'feat'), _raw(tbias_layer), tb_stride, _f(gamma, 'gamma'), _f(beta, 'beta'),
              _b16(out, 'h'), _f(mean, 'mean'), _f(rstd, 'rstd'), B, L, _stream())
    return out


def stem_ln_bf16(x, w_t, bias, tbias_layer, tb_stride, gamma, beta, feat=None, out=None, mean=None, rstd=None):
    """stem_fwd on int64 tokens and ln_fwd_bf16 of its output in one launch (drk_stem_ln_bf16): returns (feat, h)."""
    B, L = x.shape[0], x.shape[1]
    feat = feat if feat is not None else torch.empty(B, L, C, dtype=torch.float32, device=x.device)
    out = out if out is not None else torch.empty(B, L, C, dtype=torch.bfloat16, device=x.device)
    _lib.call('drk_stem_ln_bf16', _i(x, 'x'), _f(w_t, 'w_t'), _f(bias, 'bias'), _raw(tbias_layer), tb_stride, _f(gamma, 'gamma'),
              _f(beta, 'beta'), _f(feat, 'feat'), _b16(out, 'h'), _f(mean, 'mean'), _f(rstd, 'rstd'), B, L, _stream())
    return feat, out


def layer_bf16(h, wfrag, feat_out, *, feat=None, h_out=None, bias=None, tbias=None, tb_stride=0, gamma=None, beta=None,
               taps=9, dilation=1, relu=True, relu_mask=None, mean=None, rstd=None):
    """One launch of drk_layer_bf16 (see include/drakes_hip.h).  wfrag: a view into the packed fragment stream that
    starts at this layer's first tap.  ln is on when gamma is given."""
    B, L, _ = h.shape
    _lib.call('drk_layer_bf16', _b16(h, 'h'), _f(feat, 'feat'), _f(feat_out, 'feat_out'), _b16(h_out, 'h_out'),
              _b16(wfrag, 'wfrag'), _raw(bias), _raw(tbias), tb_stride, _raw(gamma), _raw(beta), B, L, taps, dilation,
              1 if gamma is not None else 0, 1 if relu else 0, _dev_ptr(relu_mask, torch.uint8, 'relu_mask'),
              _f(mean, 'mean'), _f(rstd, 'rstd'), _stream())
    return feat_out


def gate_bf16(g, relu_mask=None, y=None, colsum=None, out=None):
    """out = bf16(g * gate), gate = relu_mask (uint8) or (y > 0); colsum [128] += column sums (deterministic)."""
    B, L, _ = g.shape
    out = out if out is not None else torch.empty(B, L, C, dtype=torch.bfloat16, device=g.device)
    ws = workspace(_lib.load().drk_gate_bf16_workspace_bytes(), g.device) if colsum is not None else None
    _lib.call('drk_gate_bf16', _f(g, 'g'), _dev_ptr(relu_mask, torch.uint8, 'relu_mask'), _f(y, 'y'), _b16(out, 'out'),
              _f(colsum, 'colsum'), B, L, _f(ws, 'workspace'), ws.numel() * 4 if ws is not None else 0, _stream())
    return out


def conv_wgrad_bf16(h, dc, dwt, dilation, ws=None):
    """dwt [taps,128(ci),128(co)] += sum_rows h[row + off][ci] * dc[row][co] from bf16 operands (fp32 accumulation,
    deterministic split-K through the workspace)."""
    B, L, _ = h.shape
    taps = dwt.shape[0]
    if ws is None:
        ws = workspace(_lib.load().drk_conv_wgrad_bf16_workspace_bytes(B, L, taps), h.device)
    _lib.call('drk_conv_wgrad_bf16', _b16(h, 'h'), _b16(dc, 'dc'), _f(dwt, 'dwt'), B, L, taps, dilation, _f(ws, 'workspace'),
              ws.numel() * 4, _stream())


def layer_bwd_bf16(dc, wfrag_t, gout, *, feat=None, tbias=None, gamma=None, mean=None, rstd=None, gr=None, next_mask=None,
                   dc_next=None, partials=None, part_stride=0, taps=9, dilation=1):
    """One launch of drk_layer_bwd_bf16 (include/drakes_hip.h): the convolution's input gradient with the LayerNorm
    backward (on when gamma is given), the residual add and the gate of the layer below in its epilogue.  partials: a view
    starting at this launch's [4][128] slot of workgroup 0's slab; part_stride floats between workgroups' slabs."""
    B, L, _ = dc.shape
    _lib.call('drk_layer_bwd_bf16', _b16(dc, 'dc'), _b16(wfrag_t, 'wfrag_t'), _f(feat, 'feat'), _raw(tbias), _raw(gamma),
              _f(mean, 'mean'), _f(rstd, 'rstd'), _f(gr, 'gr'), _f(gout, 'gout'), _dev_ptr(next_mask, torch.uint8, 'next_mask'),
              _b16(dc_next, 'dc_next'), _raw(partials), part_stride, B, L, taps, dilation, 1 if gamma is not None else 0,
              _stream())
    return gout


def head_bwd_bf16(dlogits, y, w, dc, partials=None, part_stride=0):
    """dc = bf16((dlogits x final_conv.2) * (y > 0)) and the per-workgroup partial sums of final_conv.2's weight / bias and
    final_conv.0's bias gradients (drk_head_bwd_bf16)."""
    B, L, _ = y.shape
    _lib.call('drk_head_bwd_bf16', _f(dlogits, 'dlogits'), _f(y, 'y'), _f(w, 'w'), _b16(dc, 'dc'), _raw(partials), part_stride,
              B, L, _stream())
    return dc


def conv_wgrad_bf16_multi(h_ptrs, dc_ptrs, n_seg, dwt, B, L, dilation, ws=None):
    """dwt [taps,128,128] += the weight gradient summed over n_seg recorded (h, dc) pairs of shape [B,L,128] bf16, in one
    launch; h_ptrs / dc_ptrs: int64 device tensors [n_seg] of device addresses."""
    taps = dwt.shape[0]
    if ws is None:
        ws = workspace(_lib.load().drk_conv_wgrad_bf16_multi_workspace_bytes(n_seg, B, L, taps), dwt.device)
    _lib.call('drk_conv_wgrad_bf16_multi', _i(h_ptrs, 'h_ptrs'), _i(dc_ptrs, 'dc_ptrs'), n_seg, _f(dwt, 'dwt'), B, L, taps,
              dilation, _f(ws, 'workspace'), ws.numel() * 4, _stream())


DENOISER_PER_LAYER = False     # tests / A-B runs: True takes the per-layer launches instead of the fused forward


def denoiser_flags(per_layer=None):
    return 1 if (DENOISER_PER_LAYER if per_layer is None else per_layer) else 0


def denoiser_saved_buffers(B, L, n_layers, device, per_layer=None):
    """(feats float32 [.], masks uint8 [.]) sized for drk_denoiser_fwd_bf16 in the mode the flags select: per-layer launches
    keep row-major blocks ([n+1][B*L][128] floats, [n][B*L][128] gate bytes); the fused kernels keep feat0 row-major followed
    by the residual snapshots in their lane order, and the gates packed (see include/drakes_hip.h)."""
    lib, fl = _lib.load(), denoiser_flags(per_layer)
    return (torch.empty(lib.drk_denoiser_feats_floats(B, L, n_layers, fl), dtype=torch.float32, device=device),
            torch.empty(lib.drk_denoiser_masks_bytes(B, L, n_layers, fl), dtype=torch.uint8, device=device))


def denoiser_is_fused(L, n_layers, per_layer=None):
    return bool(_lib.load().drk_denoiser_is_fused(L, n_layers, denoiser_flags(per_layer)))


def denoiser_fwd_bf16(seq, pk, st, tbias, feats, hs, masks, means, rstds, y, logits, per_layer=None):
    """drk_denoiser_fwd_bf16: the whole differentiable forward in one native call.  pk = CNNModel.packed(), st =
    fastpath._packed(bb); tbias [n_layers,128] (one row for the batch); feats / masks: denoiser_saved_buffers(...) of the
    same mode; the other outputs are whole row-major blocks (see the header)."""
    B, L = seq.shape[0], seq.shape[1]
    tok = seq.dtype == torch.int64
    lib, fl, n = _lib.load(), denoiser_flags(per_layer), st['n_layers']
    if feats.numel() < lib.drk_denoiser_feats_floats(B, L, n, fl) or masks.numel() < lib.drk_denoiser_masks_bytes(B, L, n, fl):
        raise ValueError('feats / masks: too small for this mode (ops.denoiser_saved_buffers)')
    _lib.call('drk_denoiser_fwd_bf16', _i(seq, 'x') if tok else None, None if tok else _f(seq, 'x_soft'), _f(pk['stem_w'], 'stem_w'),
              _f(pk['stem_b'], 'stem_b'), _b16(st['wstream'], 'wstream'), _f(st['vecs'], 'vecs'), _f(tbias, 'tbias'),
              _f(pk['head2_w'], 'head2_w'), _f(pk['head2_b'], 'head2_b'), _f(feats, 'feats'), _b16(hs, 'hs'),
              _dev_ptr(masks, torch.uint8, 'masks'), _f(means, 'means'), _f(rstds, 'rstds'), _f(y, 'y'), _f(logits, 'logits'),
              B, L, n, st['dil'], fl, _stream())
    return logits


def denoiser_bwd_bf16(dlogits, seq, pk, st, wstream_t, tbias, feats, masks, means, rstds, y, dcs, gr, partials=None, acc=None,
                      stem_dw=None, stem_db=None, dx=None, per_layer=None):
    """drk_denoiser_bwd_bf16: the whole backward except weight gradients and the time path, in one native call; `per_layer`
    as the forward that filled feats / masks ran."""
    B, L = seq.shape[0], seq.shape[1]
    tok = seq.dtype == torch.int64
    lib, fl, n = _lib.load(), denoiser_flags(per_layer), st['n_layers']
    if feats.numel() < lib.drk_denoiser_feats_floats(B, L, n, fl) or masks.numel() < lib.drk_denoiser_masks_bytes(B, L, n, fl):
        raise ValueError('feats / masks: too small for this mode (ops.denoiser_saved_buffers)')
    _lib.call('drk_denoiser_bwd_bf16', _f(dlogits, 'dlogits'), _f(y, 'y'), _f(pk['head2_w'], 'head2_w'), _b16(wstream_t, 'wstream_t'),
              _f(st['vecs'], 'vecs'), _f(tbias, 'tbias'), _f(feats, 'feats'), _dev_ptr(masks, torch.uint8, 'masks'),
              _f(means, 'means'), _f(rstds, 'rstds'), _b16(dcs, 'dcs'), dcs.shape[0], _f(gr, 'gr'), _f(partials, 'partials'),
              _f(acc, 'acc'), _i(seq, 'x') if tok else None, None if tok else _f(seq, 'x_soft'), _f(pk['stem_w'], 'stem_w'),
              _f(stem_dw, 'stem_dw'), _f(stem_db, 'stem_db'), _f(dx, 'dx'), B, L, n, st['dil'], fl, _stream())


def stem_wgrad_multi(dpre_ptrs, x_ptrs, tokens, n_seg, dw_t, db, B, L):
    """drk_stem_wgrad_multi: dw_t [9,5,128] += and db [1,128] += over n_seg recorded calls; dpre_ptrs / x_ptrs: int64
    device tensors of n_seg addresses (x: token tensors when `tokens`, else soft [B,L,5] tensors)."""
    xp = _i(x_ptrs, 'x_ptrs')
    _lib.call('drk_stem_wgrad_multi', _i(dpre_ptrs, 'dpre_ptrs'), xp if tokens else None, None if tokens else xp, n_seg,
              _f(dw_t, 'dw_t'), _f(db, 'db'), B, L, _stream())


def denoiser_bwd_slabs(B, L):
    """Partial slabs drk_denoiser_bwd_bf16 wants in its workspace."""
    return int(_lib.load().drk_denoiser_bwd_bf16_slabs(B, L))


def slab_reduce(slabs, n_slabs, stride, out, n):
    """out[:n] += slabs[0][:n] + slabs[1][:n] + ... in order (slab s at element s * stride of `slabs`)."""
    _lib.call('drk_slab_reduce', _raw(slabs), n_slabs, stride, _raw(out), n, _stream())
    return out


# ---- weight layouts the kernels read (pure index permutations) --------------------------------
def pack_conv_weight_f32(w):
    """[co][ci][taps] -> [taps][ci/4][co][4] (drk_conv_fwd_f32's w_pk)."""
    co, ci, taps = w.shape
    return w.detach().permute(2, 1, 0).reshape(taps, ci // 4, 4, co).permute(0, 1, 3, 2).contiguous()


def pack_stem_weight(w):
    """[co][5][9] -> [9][5][co]."""
    return w.detach().permute(2, 1, 0).contiguous()


# ---- fine-tune loss: per-step KL term ----------------------------------------------------------
def _flag_ptrs(flag):
    """copy_flag [B,L] or [B,L,1]: float32 (soft steps) or int64 (hard steps) -> (float ptr, int64 ptr)."""
    if flag.dtype == torch.int64:
        return None, _i(flag, 'copy_flag')
    return _f(flag, 'copy_flag'), None


def kl_step_fwd(logp, logp_old, last_x, flag, mc_t):
    B, L, _ = logp.shape
    kl = torch.empty(B, dtype=torch.float32, device=logp.device)
    ff, fi = _flag_ptrs(flag)
    _lib.call('drk_kl_step_fwd', _f(logp, 'logp'), _f(logp_old, 'logp_old'), _f(last_x, 'last_x'), ff, fi, float(mc_t),
              _f(kl, 'kl'), B, L, _stream())
    return kl


def kl_step_bwd(logp, logp_old, last_x, flag, grad_kl, mc_t, want_dold, want_dx, want_dflag):
    B, L, _ = logp.shape
    dlogp = torch.empty_like(logp)
    dold = torch.empty_like(logp) if want_dold else None
    dx = torch.empty_like(last_x) if want_dx else None
    dflag = torch.empty(B, L, dtype=torch.float32, device=logp.device) if want_dflag else None
    ff, fi = _flag_ptrs(flag)
    _lib.call('drk_kl_step_bwd', _f(logp, 'logp'), _f(logp_old, 'logp_old'), _f(last_x, 'last_x'), ff, fi,
              _f(grad_kl, 'grad_kl'), float(mc_t), _f(dlogp, 'dlogp'), _f(dold, 'dlogp_old'), _f(dx, 'dlast_x'),
              _f(dflag, 'dflag'), B, L, _stream())
    return dlogp, dold, dx, dflag


# ---- likelihood ----------------------------------------------------------------------------------
def q_xt(x0, move_chance, seed=0, step=0, sample_offset=0, uniforms=None, out=None, mc_rows=None):
    """mc_rows: optional float32 [B,1] of per-row move chances; then move_chance is ignored."""
    B, L = x0.shape
    xt = out if out is not None else torch.empty_like(x0)
    _lib.call('drk_q_xt', _i(x0, 'x0'), _f(uniforms, 'uniforms'), _i(xt, 'xt'), B, L, float(move_chance),
              _mc_rows(mc_rows, B, 1), seed, step, sample_offset, _stream())
    return xt


def likelihood_accum(logits, x0, xt, multiplier, acc):
    B, L = x0.shape
    _lib.call('drk_likelihood_accum', _f(logits, 'logits'), _i(x0, 'x0'), _i(xt, 'xt'), float(multiplier), _f(acc, 'acc'),
              B, L, _stream())
    return acc


def reverse_step_hard_table(logits, x, mc_table, step_counter, rng, out):
    """Graph-replayable hard step: per-step scalars, the step index and (seed, sample_offset) come from device memory."""
    B, L = x.shape
    if step_counter.dtype != torch.int32 or rng.dtype != torch.int64 or rng.numel() != 2:
        raise TypeError('step_counter: int32 [1]; rng: int64 [2] (the bit patterns of two uint64)')
    _lib.call('drk_reverse_step_hard_table', _f(logits, 'logits'), _i(x, 'x'), _i(out, 'x_next'), B, L,
              _f(mc_table, 'mc_table'), ctypes.c_void_p(step_counter.data_ptr()), _i(rng, 'rng'), _stream())
    return out


# ---- wide vocabulary (protein, V = 22) -------------------------------------------------------------
def reverse_step_hard_wide(logits, x, d_t, move_chance_s, temp=1.0, mask_index=21, seed=0, step=0, sample_offset=0,
                           uniforms=None, out=None, want_clean=True):
    """One step of the protein sampler (fm_utils.py:113-134).  Returns (x_next, pred_clean or None)."""
    B, L, Vw = logits.shape
    x_next = out if out is not None else torch.empty_like(x)
    clean = torch.empty_like(x) if want_clean else None
    _lib.call('drk_reverse_step_hard_wide', _f(logits, 'logits'), _i(x, 'x'), _f(uniforms, 'uniforms'), _i(x_next, 'x_next'),
              _i(clean, 'pred_clean'), B, L, Vw, mask_index, float(temp), float(d_t), float(move_chance_s), seed, step,
              sample_offset, _stream())
    return x_next, clean


def reverse_step_soft_wide_fwd(logits, x_soft, d_t, move_chance_s, temp, gumbel_temp, mask_index=21, seed=0, step=0,
                               sample_offset=0, uniforms=None):
    B, L, Vw = x_soft.shape
    x_next = torch.empty_like(x_soft)
    flag = torch.empty(B, L, dtype=torch.float32, device=x_soft.device)
    _lib.call('drk_reverse_step_soft_wide_fwd', _f(logits, 'logits'), _f(x_soft, 'x_soft'), _f(uniforms, 'uniforms'),
              _f(x_next, 'x_next'), _f(flag, 'flag'), B, L, Vw, mask_index, float(temp), float(d_t), float(move_chance_s),
              float(gumbel_temp), seed, step, sample_offset, _stream())
    return x_next, flag


def reverse_step_soft_wide_bwd(logits, x_soft, grad_x_next, grad_flag, d_t, move_chance_s, temp, gumbel_temp, mask_index=21,
                               seed=0, step=0, sample_offset=0, uniforms=None):
    B, L, Vw = x_soft.shape
    dlogits, dx = torch.empty_like(logits), torch.empty_like(x_soft)
    _lib.call('drk_reverse_step_soft_wide_bwd', _f(logits, 'logits'), _f(x_soft, 'x_soft'), _f(uniforms, 'uniforms'),
              _f(grad_x_next, 'grad_x_next'), _f(grad_flag, 'grad_flag'), _f(dlogits, 'dlogits'), _f(dx, 'dx'), B, L, Vw,
              mask_index, float(temp), float(d_t), float(move_chance_s), float(gumbel_temp), seed, step, sample_offset,
              _stream())
    return dlogits, dx
