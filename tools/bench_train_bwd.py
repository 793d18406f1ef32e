"""Times one differentiable denoiser backward (drk_denoiser_bwd_bf16): the fused launch against the per-layer launches.

    [DRK_LIB=drakes_amd/libexp_X.so] python tools/bench_train_bwd.py [B ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import drakes_amd  # noqa: E402
from drakes_amd import autograd as AG, fastpath, ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    L = 200
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=L)).to(dev)
    bb = m.backbone
    pk, st = bb.packed(), fastpath._packed(bb)
    wT = fastpath._packed_T(bb)
    n = bb.num_layers
    W = int(AG.GradSink(bb).W)
    tbias = bb.time_bias(torch.zeros(1, device=dev))[0]
    for B in [int(v) for v in sys.argv[1:]] or [128, 32, 256]:
        x = torch.softmax(torch.randn(B, L, 5, device=dev), -1)
        dl = torch.randn(B, L, 5, device=dev)
        dcs = torch.empty(n + 1, B, L, 128, dtype=torch.bfloat16, device=dev)
        gr = torch.empty(B, L, 128, device=dev)
        dx = torch.empty(B, L, 5, device=dev)
        kw = dict(partials=torch.empty(ops.denoiser_bwd_slabs(B, L) * W, device=dev), acc=torch.zeros(W, device=dev),
                  stem_dw=torch.zeros(9, 5, 128, device=dev), stem_db=torch.zeros(1, 128, device=dev))
        for per_layer in (False, True):
            ops.DENOISER_PER_LAYER = per_layer
            feats, masks = ops.denoiser_saved_buffers(B, L, n, dev, per_layer)
            fw = dict(feats=feats, masks=masks, hs=torch.empty(n + 1, B, L, 128, dtype=torch.bfloat16, device=dev),
                      means=torch.empty(n, B * L, device=dev), rstds=torch.empty(n, B * L, device=dev),
                      y=torch.empty(B, L, 128, device=dev), logits=torch.empty(B, L, 5, device=dev))
            ops.denoiser_fwd_bf16(x, pk, st, tbias, **fw)
            run = lambda: ops.denoiser_bwd_bf16(dl, x, pk, st, wT, tbias, fw['feats'], fw['masks'], fw['means'], fw['rstds'], fw['y'],
                                                dcs, gr, dx=dx, **kw)
            for _ in range(3):
                run()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in evs:
                a.record()
                run()
                b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            print(f'LIB={os.path.basename(os.environ.get("DRK_LIB", "default"))} B={B} backward {"per-layer" if per_layer else "fused    "}: '
                  f'median {ts[len(ts) // 2] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us', flush=True)
        ops.DENOISER_PER_LAYER = False


if __name__ == '__main__':
    main()
