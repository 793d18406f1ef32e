"""BASELINE.json configs[2]: one DRAKES fine-tune sampling pass — `_sample_finetune_gradient` (78 hard + 50 soft
steps, truncated backprop through the last K=50) + backward of a synthetic linear reward, batch 128, L=200.

    python tools/bench_finetune.py [--batch 128] [--trunc 50] [--reps 3]

Reports seconds per forward, per backward, and peak device memory (SURVEY.md §8d "C3 input").
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import drakes_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--trunc', type=int, default=50)
    ap.add_argument('--steps', type=int, default=128)
    ap.add_argument('--length', type=int, default=200)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = drakes_amd.default_config(length=a.length, steps=a.steps, truncate_steps=a.trunc)
    m = drakes_amd.Diffusion(cfg).to(dev).train()
    rw = torch.randn(a.batch, 4, a.length, device=dev)
    fwd, bwd = [], []
    for r in range(a.reps + 1):
        m.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m._sample_finetune_gradient(eval_sp_size=a.batch, seed=100 + r)[0]
        loss = -(out.permute(0, 2, 1) * rw).sum((1, 2)).mean()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if r:                                   # first pass is warm-up
            fwd.append(t1 - t0)
            bwd.append(t2 - t1)
        print(f'rep {r}: forward {t1 - t0:.3f} s, backward {t2 - t1:.3f} s', flush=True)
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
    res = {'workload': f'configs[2]: _sample_finetune_gradient + backward, batch {a.batch}, L={a.length}, {a.steps} steps, '
                       f'K={a.trunc}, fp32 kernels', 'forward_s': sorted(fwd)[len(fwd) // 2], 'backward_s': sorted(bwd)[len(bwd) // 2],
           'total_s': sorted(f + b for f, b in zip(fwd, bwd))[len(fwd) // 2],
           'peak_mem_GiB': torch.cuda.max_memory_allocated() / 2 ** 30, 'grad_norm': gn}
    print(json.dumps(res), flush=True)
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
