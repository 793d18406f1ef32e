"""Per-dilation timing of drk_layer_bf16 (one residual layer per launch, bf16 MFMA).

    python tools/bench_layer.py [--batch 256] [--length 1024] [--reps 20]
"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import drakes_amd  # noqa: E402
from drakes_amd import _lib, fastpath, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--length', type=int, default=1024)
ap.add_argument('--reps', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = drakes_amd.Diffusion(drakes_amd.default_config(length=a.length)).to(dev).eval()
bb = m.backbone
st = fastpath._packed(bb)
B, L = a.batch, a.length
feat = torch.randn(B, L, 128, device=dev)
h0 = torch.randn(B, L, 128, device=dev).to(torch.bfloat16)
h1 = torch.empty_like(h0)
tbias = bb.time_bias(torch.zeros(1, device=dev))
vec = st['vecs']
P = lambda t: ctypes.c_void_p(t.data_ptr())
flop = 2.0 * B * L * 9 * 128 * 128
byts = B * L * (256 + 512 + 512 + 256)
for i, d in ((0, 1), (2, 4), (3, 16), (4, 64)):
    args = (P(h0), P(feat), P(feat), P(h1), P(st['wstream'][9 * i:]), P(vec[3 * (i + 1)]), P(tbias[0, i + 1]), 0,
            P(vec[3 * (i + 1) + 1]), P(vec[3 * (i + 1) + 2]), B, L, 9, d, 1, 1, None, None, None, ops._stream())
    for _ in range(3):
        _lib.call('drk_layer_bf16', *args)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for s, e in evs:
        s.record()
        _lib.call('drk_layer_bf16', *args)
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    t = ts[len(ts) // 2]
    print(f'dilation {d:2d}: {t * 1e3:7.1f} us  {flop / t / 1e9:7.1f} TFLOP/s  {byts / t / 1e6:6.0f} GB/s algorithmic', flush=True)
