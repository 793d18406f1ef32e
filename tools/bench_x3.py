"""drk_conv_fwd_x3 (fp32 grade on bf16 MFMA, three planes) against drk_conv_fwd_f32: error vs float64, time per launch.

    python tools/bench_x3.py [--batch 1024] [--length 200]
"""
import argparse, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from drakes_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1024)
ap.add_argument('--length', type=int, default=200)
a = ap.parse_args()
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
# accuracy on a small case, all dilations
for (B, L, d, taps) in ((3, 200, 1, 9), (2, 77, 4, 9), (2, 200, 16, 9), (3, 200, 64, 9), (1, 1024, 64, 9), (5, 33, 0, 1)):
    w = torch.randn(128, 128, taps, generator=g) * 0.05
    h = torch.randn(B, L, 128, generator=g)
    bias = torch.randn(128, generator=g) * 0.1
    resid = torch.randn(B, L, 128, generator=g)
    ref = F.conv1d(h.double().permute(0, 2, 1), w.double(), bias.double(), dilation=max(d, 1), padding=(taps // 2) * max(d, 1)).permute(0, 2, 1)
    want = torch.relu(ref) + resid.double()
    hd, rd, bd = h.to(dev), resid.to(dev), bias.to(dev)
    m1 = torch.empty(B, L, 128, dtype=torch.uint8, device=dev)
    m2 = torch.empty_like(m1)
    o1 = ops.conv_fwd_f32(hd, ops.pack_conv_weight_f32(w.to(dev)), bd, rd, d, relu=True, relu_mask=m1)
    o2 = ops.conv_fwd_x3(hd, ops.pack_conv_weight_x3(w.to(dev)), bd, rd, d, relu=True, relu_mask=m2)
    e1 = (o1.double().cpu() - want).abs().max().item()
    e2 = (o2.double().cpu() - want).abs().max().item()
    print(f'B={B} L={L} d={d} taps={taps}: max err f32 kernel {e1:.2e}, x3 kernel {e2:.2e} (|out| max {want.abs().max():.1f}); '
          f'mask differs at {(m1 != m2).float().mean().item():.2e}', flush=True)
B, L = a.batch, a.length
w = torch.randn(128, 128, 9, generator=g) * 0.05
h = torch.randn(B, L, 128, device=dev)
feat = torch.randn(B, L, 128, device=dev)
bias = torch.zeros(128, device=dev)
wf, w3 = ops.pack_conv_weight_f32(w.to(dev)), ops.pack_conv_weight_x3(w.to(dev))
flop = 2.0 * B * L * 9 * 128 * 128
for d in (1, 4, 16, 64):
    for name, fn, wp in (('f32', ops.conv_fwd_f32, wf), ('x3', ops.conv_fwd_x3, w3)):
        for _ in range(3):
            fn(h, wp, bias, feat, d, relu=True, out=feat)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for s, e in evs:
            s.record(); fn(h, wp, bias, feat, d, relu=True, out=feat); e.record()
        torch.cuda.synchronize()
        t = sorted(s.elapsed_time(e) for s, e in evs)[5]
        print(f'dilation {d:2d} {name:4s}: {t * 1e3:8.1f} us  {flop / t / 1e9:7.1f} TFLOP/s fp32-equivalent', flush=True)
