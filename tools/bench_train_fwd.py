"""Times one differentiable denoiser forward (drk_denoiser_fwd_bf16): the fused launch against the per-layer launches.

    [DRK_LIB=drakes_amd/libexp_X.so] python tools/bench_train_fwd.py [B ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import drakes_amd  # noqa: E402
from drakes_amd import fastpath, ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    L = 200
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=L)).to(dev)
    bb = m.backbone
    pk, st = bb.packed(), fastpath._packed(bb)
    n = bb.num_layers
    tbias = bb.time_bias(torch.zeros(1, device=dev))[0]
    for B in [int(v) for v in sys.argv[1:]] or [128, 32, 256]:
        x = torch.softmax(torch.randn(B, L, 5, device=dev), -1)
        for per_layer in (False, True):
            feats, masks = ops.denoiser_saved_buffers(B, L, n, dev, per_layer)
            blocks = dict(feats=feats, masks=masks, hs=torch.empty(n + 1, B, L, 128, dtype=torch.bfloat16, device=dev),
                          means=torch.empty(n, B * L, device=dev), rstds=torch.empty(n, B * L, device=dev),
                          y=torch.empty(B, L, 128, device=dev), logits=torch.empty(B, L, 5, device=dev))
            ops.DENOISER_PER_LAYER = per_layer
            for _ in range(3):
                ops.denoiser_fwd_bf16(x, pk, st, tbias, **blocks)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in evs:
                a.record()
                ops.denoiser_fwd_bf16(x, pk, st, tbias, **blocks)
                b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            print(f'LIB={os.path.basename(os.environ.get("DRK_LIB", "default"))} B={B} {"per-layer" if per_layer else "fused    "}: '
                  f'median {ts[len(ts) // 2] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us', flush=True)
        ops.DENOISER_PER_LAYER = False


if __name__ == '__main__':
    main()
