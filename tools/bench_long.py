"""BASELINE.json configs[4] on ONE GPU: 1024-bp sequences, 512 steps; fp32 kernels or (--dtype bf16) the
layer-per-launch bf16 kernels (the fused bf16 loop holds L <= 224).

    python tools/bench_long.py [--batch 64] [--steps 512] [--reps 2] [--dtype fp32|bf16] [--profile]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import drakes_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--steps', type=int, default=512)
ap.add_argument('--length', type=int, default=1024)
ap.add_argument('--reps', type=int, default=2)
ap.add_argument('--json', default=None)
ap.add_argument('--dtype', default='fp32')
ap.add_argument('--profile', action='store_true', help='per-kernel HIP-event times of the last repetition')
a = ap.parse_args()
torch.manual_seed(0)
m = drakes_amd.Diffusion(drakes_amd.default_config(length=a.length, steps=a.steps)).to('cuda:0').eval()
m.hip_dtype = a.dtype
from drakes_amd import ops  # noqa: E402
ts = []
for r in range(a.reps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if a.profile and r == a.reps:
        ops.PROFILE.enable(every=7)
    x = m._sample(eval_sp_size=a.batch, seed=r)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'rep {r}: {ts[-1]:.3f} s', flush=True)
t = sorted(ts[1:])[len(ts[1:]) // 2]
flop = 2.0 * a.batch * a.length * (45 * 128 + 20 * 9 * 128 * 128 + 128 * 128 + 128 * 5) * (a.steps + 1)
res = {'workload': f'configs[4] on one GPU: _sample, L={a.length}, {a.steps} steps, batch {a.batch}, {a.dtype} kernels',
       'seconds': t, 'sequences_per_sec': a.batch / t, 'TFLOPs_dense': flop / t / 1e12}
if a.profile:
    name, n, ms, fl = ops.PROFILE.disable()
    res['kernel'] = {'name': name, 'launches_timed': n, 'avg_ms': ms, 'TFLOPs': fl / ms / 1e9 if ms else None}
print(json.dumps(res), flush=True)
if a.json:
    json.dump(res, open(a.json, 'w'), indent=1)
