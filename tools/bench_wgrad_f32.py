"""Time of one drk_conv_wgrad_f32 call (k_conv_wgrad + the ordered slab reduction) at fine-tune and sampling batch sizes.

    python tools/bench_wgrad_f32.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from drakes_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
for B, L in ((128, 200), (1024, 200)):
    h = torch.randn(B, L, 128, device=dev)
    dc = torch.randn(B, L, 128, device=dev)
    for d in (1, 16, 64):
        dwt = torch.zeros(9, 128, 128, device=dev)
        db = torch.zeros(128, device=dev)
        for _ in range(3):
            ops.conv_wgrad_f32(h, dc, dwt, d, db)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for s, e in evs:
            s.record(); ops.conv_wgrad_f32(h, dc, dwt, d, db); e.record()
        torch.cuda.synchronize()
        t = sorted(s.elapsed_time(e) for s, e in evs)[10]
        print(f'wgrad f32 B={B} L={L} dilation {d:2d}: {t * 1e3:7.1f} us  {2.0 * B * L * 9 * 128 * 128 / t / 1e9:6.1f} TFLOP/s', flush=True)
