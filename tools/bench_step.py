"""Stand-alone HBM roofline of the reverse-step kernels (SURVEY.md §8d "Roofline, reverse step").

    python tools/bench_step.py [--positions 16777216] [--reps 20] [--json gpurun_out/step_roofline.json]

Times each kernel with HIP events on the launch stream at a batch large enough to be bandwidth-bound and reports
ALGORITHMIC bytes / time against the 8 TB/s HBM3E peak.  Algorithmic bytes per position (fp32 logits, int64 tokens,
as the reference's dtypes): hard step 36 B with in-kernel Philox, 56 B with injected uniforms; all positions masked,
so every logit row is really read.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from drakes_amd import _lib, ops  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], sum(ts) / len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--positions', type=int, default=1 << 24)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    L = 200
    B = a.positions // L
    n = B * L
    g = torch.Generator(device=dev).manual_seed(0)
    logits = torch.randn(B, L, 5, device=dev, generator=g)
    u = torch.rand(B, L, 5, device=dev, generator=g)
    x_mask = torch.full((B, L), 4, dtype=torch.int64, device=dev)
    x_half = torch.where(torch.rand(B, L, device=dev, generator=g) < 0.5, x_mask,
                         torch.randint(0, 4, (B, L), device=dev, generator=g))
    x_soft = torch.softmax(torch.randn(B, L, 5, device=dev, generator=g), -1)
    out_tok = torch.empty_like(x_mask)
    o5a, o5b = torch.empty_like(logits), torch.empty_like(logits)
    flag = torch.empty(B, L, device=dev)
    gflag = torch.randn(B, L, device=dev, generator=g)
    s = ops._stream()
    P = lambda t: t.data_ptr()
    mc_t, mc_s = 0.49950498, 0.49170041
    mc_rows = torch.tensor([[mc_t, mc_s]], device=dev).repeat(B, 1).contiguous()      # per-row timesteps (round 2)

    cases = [
        ('reverse_step_hard, Philox in-kernel, all masked', 36,
         lambda: _lib.call('drk_reverse_step_hard', P(logits), P(x_mask), None, P(out_tok), B, L, mc_t, mc_s, None, 7, 3, 0, s)),
        ('reverse_step_hard, Philox in-kernel, half masked', 36,
         lambda: _lib.call('drk_reverse_step_hard', P(logits), P(x_half), None, P(out_tok), B, L, mc_t, mc_s, None, 7, 3, 0, s)),
        ('reverse_step_hard, Philox in-kernel, all masked, per-row mc', 36,
         lambda: _lib.call('drk_reverse_step_hard', P(logits), P(x_mask), None, P(out_tok), B, L, mc_t, mc_s, P(mc_rows), 7, 3, 0, s)),
        ('reverse_step_hard, injected uniforms, all masked', 56,
         lambda: _lib.call('drk_reverse_step_hard', P(logits), P(x_mask), P(u), P(out_tok), B, L, mc_t, mc_s, None, 7, 3, 0, s)),
        ('subs_fwd (tokens)', 48,
         lambda: _lib.call('drk_subs_fwd', P(logits), P(x_half), None, P(o5a), B, L, s)),
        ('denoise_argmax', 36,
         lambda: _lib.call('drk_denoise_argmax', P(logits), P(x_half), P(out_tok), B, L, s)),
        ('step_uniforms (Philox only)', 20,
         lambda: _lib.call('drk_step_uniforms', P(o5a), B, L, 7, 3, 0, s)),
        ('reverse_step_soft_fwd, Philox in-kernel', 20 + 20 + 20 + 4,
         lambda: _lib.call('drk_reverse_step_soft_fwd', P(logits), P(x_soft), None, P(o5a), P(flag), B, L, mc_t, mc_s,
                           None, 1.0, 0.0, 7, 3, 0, s)),
        ('reverse_step_soft_bwd, Philox in-kernel', 20 + 20 + 20 + 4 + 20 + 20,
         lambda: _lib.call('drk_reverse_step_soft_bwd', P(logits), P(x_soft), None, P(u), P(gflag), P(o5a), P(o5b), B, L,
                           mc_t, mc_s, None, 1.0, 0.0, 7, 3, 0, s)),
    ]
    rows = []
    for name, bpp, fn in cases:
        med, avg = timed(fn, a.reps)
        gbs = n * bpp / (med * 1e-3) / 1e9
        rows.append({'kernel': name, 'positions': n, 'bytes_per_position': bpp, 'median_ms': med, 'avg_ms': avg,
                     'achieved_GBs': gbs, 'frac_of_8TBs': gbs / HBM_PEAK_GBS})
        print(f'{name:52s} {bpp:4d} B/pos  {med:8.4f} ms  {gbs:8.1f} GB/s  {100 * gbs / HBM_PEAK_GBS:5.1f} % of HBM peak',
              flush=True)
    # the in-sampler cost at B=1024 (launch-bound): same kernel, 204 800 positions
    Bs = 1024
    med, _ = timed(lambda: _lib.call('drk_reverse_step_hard', P(logits), P(x_mask), None, P(out_tok), Bs, L, mc_t, mc_s, None, 7,
                                     3, 0, s), 50)
    print(f'in-sampler size B=1024: {med * 1e3:.1f} us per step ({Bs * L * 36 / (med * 1e-3) / 1e9:.0f} GB/s)', flush=True)
    rows.append({'kernel': 'reverse_step_hard at B=1024 (launch-bound)', 'positions': Bs * L, 'median_ms': med})
    if a.json:
        with open(a.json, 'w') as f:
            json.dump({'drk_version': _lib.load().drk_version(), 'rows': rows}, f, indent=1)


if __name__ == '__main__':
    main()
