"""Time of one drk_ln_bwd launch (LayerNorm backward with the fused hand-over to the layer below) at fine-tune shapes.

    python tools/bench_ln_bwd.py [--batch 128] [--length 200]
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from drakes_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=128)
ap.add_argument('--length', type=int, default=200)
a = ap.parse_args()
dev = torch.device('cuda:0')
B, L = a.batch, a.length
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev)
dh, feat, g_resid = r(B, L, 128), r(B, L, 128), r(B, L, 128)
gamma, tb = r(128), r(128)
mean, rstd = r(B * L), r(B * L).abs() + 0.5
dfeat = torch.empty(B, L, 128, device=dev)
mask = (torch.rand(B, L, 128, device=dev) > 0.5).to(torch.uint8)
dc = torch.empty(B, L, 128, dtype=torch.bfloat16, device=dev)
for fused in (False, True):
    acc = [torch.zeros(128, device=dev) for _ in range(4)]
    kw = dict(next_mask=mask, dc_next=dc, dbias_next=acc[3]) if fused else {}
    call = lambda: ops.ln_bwd(dh, feat, tb, 0, gamma, mean, rstd, g_resid, None, dfeat, acc[0], acc[1], acc[2], **kw)
    for _ in range(3):
        call()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for s, e in evs:
        s.record(); call(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)[15]
    byts = B * L * (3 * 512 + 512 + 8 + (128 + 256 if fused else 0))
    print(f'ln_bwd B={B} L={L} fused hand-over={fused}: {t * 1e3:6.1f} us  {byts / t / 1e6:6.0f} GB/s', flush=True)
