"""The allocation-free weight-gradient entry point, drk_conv_wgrad_f32_static (no counterpart in the reference: for
callers that cannot hipMalloc).  Through the C ABI, against drk_conv_wgrad_f32 (the workspace form): same chunking, same
summation order, so the bar is BIT equality; and against a float64 torch reference at the fp32 gradient tolerance.
Two calls in a row, the second smaller (the scratch keeps the first call's slabs), then a larger one again, the head's
1-tap form, and the size limit."""
import ctypes

import numpy as np
import pytest
import torch

from drakes_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _reference(h, dc, taps, dil):
    """dwt[t][ci][co] = sum_{b,l} h[b, l + (t - taps//2) * dil, ci] * dc[b, l, co], float64 on the host."""
    B, L, C = h.shape
    h64, d64 = h.double().cpu(), dc.double().cpu()
    out = torch.zeros(taps, C, C, dtype=torch.float64)
    for t in range(taps):
        off = (t - taps // 2) * dil
        lo, hi = max(0, -off), min(L, L - off)
        if hi > lo:
            out[t] = torch.einsum('blc,bld->cd', h64[:, lo + off:hi + off], d64[:, lo:hi])
    return out, d64.sum((0, 1))


def _both(B, L, taps, dil, seed, dev):
    g = torch.Generator(device='cpu').manual_seed(seed)
    h = torch.randn(B, L, 128, generator=g).to(dev)
    dc = torch.randn(B, L, 128, generator=g).to(dev)
    out = []
    for fn in (ops.conv_wgrad_f32, ops.conv_wgrad_f32_static):
        dwt = torch.full((taps, 128, 128), 0.5, device=dev)          # accumulate semantics: starts non-zero
        db = torch.full((128,), -0.25, device=dev)
        fn(h, dc, dwt, dil, db)
        out.append((dwt, db))
    ref_w, ref_b = _reference(h, dc, taps, dil)
    return out, ref_w, ref_b


def test_static_wgrad_matches_workspace_form(dev):
    cases = [(16, 200, 9, 4, 1),      # 3200 rows
             (3, 77, 9, 16, 2),       # SMALLER second call: 231 rows, ragged last chunk, taps partly out of range
             (128, 200, 9, 1, 3),     # configs[2] size: 25 600 rows
             (5, 200, 1, 0, 4),       # final_conv.0: one tap
             (2, 40, 9, 64, 5)]       # dilation beyond the sequence: only the centre tap sees data
    for B, L, taps, dil, seed in cases:
        (ws_form, st_form), ref_w, ref_b = _both(B, L, taps, dil, seed, dev)
        assert torch.equal(st_form[0], ws_form[0]), (B, L, taps, dil)
        assert torch.equal(st_form[1], ws_form[1]), (B, L, taps, dil)
        got_w = st_form[0].double().cpu() - 0.5
        got_b = st_form[1].double().cpu() + 0.25
        scale = float(ref_w.abs().max())
        assert float((got_w - ref_w).abs().max()) <= 2e-5 * scale + 1e-4, (B, L, taps, dil)
        assert float((got_b - ref_b).abs().max()) <= 2e-5 * float(ref_b.abs().max()) + 1e-4


def test_static_wgrad_rejects_what_it_cannot_hold(dev):
    h = torch.zeros(1, 8, 128, device=dev)
    dwt = torch.zeros(11, 128, 128, device=dev)
    with pytest.raises(_lib.DrakesHipError, match='beyond the built-in scratch'):
        ops.conv_wgrad_f32_static(h, h, dwt, 1)
    lib = _lib.load()
    P = ctypes.c_void_p
    rc = lib.drk_conv_wgrad_f32_static(P(h.data_ptr()), P(h.data_ptr()), P(dwt.data_ptr()), None, 1 << 15, 1 << 8, 9, 1, None)
    assert rc != 0 and b'beyond the built-in scratch' in lib.drk_last_error()      # 8 388 608 rows > the limit; nothing launched
    assert lib.drk_conv_wgrad_f32_static(None, None, None, None, 0, 8, 9, 1, None) == 0          # empty batch: no-op
