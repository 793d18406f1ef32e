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


def loop(a):
    from types import SimpleNamespace as NS
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = drakes_amd.default_config(length=a.length, steps=a.steps, truncate_steps=a.trunc)
    new_model, old_model = drakes_amd.Diffusion(cfg).to(dev), drakes_amd.Diffusion(cfg).to(dev)
    new_model.hip_dtype = old_model.hip_dtype = a.dtype
    old_model.load_state_dict(new_model.state_dict())
    reward = torch.nn.Sequential(torch.nn.Conv1d(4, 8, 5, padding=2), torch.nn.Tanh(), torch.nn.AdaptiveAvgPool1d(1),
                                 torch.nn.Flatten(), torch.nn.Linear(8, 3), torch.nn.Unflatten(1, (3, 1))).to(dev)
    for p in reward.parameters():
        p.requires_grad_(False)
    args = NS(truncate_steps=a.trunc, gumbel_temp=1.0, total_num_steps=a.steps, learning_rate=1e-4, num_epochs=1,
              num_accum_steps=a.accum, batch_size=a.batch, copy_flag_temp=None, truncate_kl=False, alpha=0.001,
              alpha_schedule_warmup=0, gradnorm_clip=1.0, save_every_n_epochs=10 ** 9)
    ts = []
    for r in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drakes_amd.fine_tune(new_model, reward, reward, old_model, args, verbose=False, **({'kl_streams': a.kl_streams} if a.kl_streams is not None else {}))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        print(f'rep {r}: {ts[-1]:.3f} s per optimizer step', flush=True)
    res = {'workload': f'fine_tune optimizer step: _sample_finetune_gradient (K={a.trunc}) + reward + KL over all {a.steps} '
                       f'steps (new + frozen model) + backward + clip + Adam, batch {a.batch} x {a.accum} micro-steps, L={a.length}, {a.dtype} kernels',
           'seconds_per_step': sorted(ts[1:])[len(ts[1:]) // 2], 'peak_mem_GiB': torch.cuda.max_memory_allocated() / 2 ** 30}
    print(json.dumps(res), flush=True)
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(res, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--trunc', type=int, default=50)
    ap.add_argument('--steps', type=int, default=128)
    ap.add_argument('--length', type=int, default=200)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--json', default=None)
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16x3', 'bf16'], help="hip_dtype: 'bf16' = fused hard phase + bf16 soft steps and backward; 'bf16x3' = fp32-grade convolutions on bf16 MFMA")
    ap.add_argument('--loop', action='store_true', help='time whole fine_tune optimizer steps (sampling + reward + KL over all steps + Adam)')
    ap.add_argument('--accum', type=int, default=1, help='--loop: micro-steps per optimizer step (the reference default is 4)')
    ap.add_argument('--kl-streams', type=int, default=None, help='fine_tune(kl_streams=...): side streams of the frozen model (0: none)')
    a = ap.parse_args()
    if a.loop:
        return loop(a)
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = drakes_amd.default_config(length=a.length, steps=a.steps, truncate_steps=a.trunc)
    m = drakes_amd.Diffusion(cfg).to(dev).train()
    m.hip_dtype = a.dtype
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
                       f'K={a.trunc}, {a.dtype} kernels', 'forward_s': sorted(fwd)[len(fwd) // 2], 'backward_s': sorted(bwd)[len(bwd) // 2],
           'total_s': sorted(f + b for f, b in zip(fwd, bwd))[len(fwd) // 2],
           'peak_mem_GiB': torch.cuda.max_memory_allocated() / 2 ** 30, 'grad_norm': gn}
    print(json.dumps(res), flush=True)
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
