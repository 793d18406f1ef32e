"""bench.py — throughput of the reverse-sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--dtype bf16|fp32] [--batch 1024]

One "step" = one full `Diffusion._sample` call (128 reverse steps + the noise-removal forward
= 129 denoiser forwards) over a batch of 1024 200-bp sequences per GPU: BASELINE.json config 2.
Weights are random-init (default torch init of the reference architecture), data is synthetic
(the sampler starts from the all-mask state), RNG is the in-kernel Philox stream.

N > 1: launched by torch.distributed.run, one rank per GPU; each rank samples its own shard of
the global batch (rows keyed by GLOBAL sample index, no data-path collective) -> weak scaling.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the
timed region) and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import drakes_amd  # noqa: E402
from drakes_amd import ops  # noqa: E402

L = 200
NUM_STEPS = 128
FORWARDS = NUM_STEPS + 1
# Dense FLOP count as the reference computes it
FLOP_PER_POS_LAYER = 2 * 128 * 128 * 9            # 294 912
FLOP_PER_POS_FWD = 11520 + 20 * FLOP_PER_POS_LAYER + 34048   # 5.944 MFLOP
PEAK = {'bf16': 2500.0, 'fp32': 157.3}            # TFLOP/s dense MFMA, MI355X_MICROARCH.md
# HBM-side bytes of ONE k_sample_bf16 launch at exactly this workload (B=1024, L=200, 129 forwards), from rocprofv3 PMC
# passes (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes
# for 16-B-per-lane loads on gfx950): tools/pmc_traffic.sh, summary in profiles/r1_bf16_v8_pmc_traffic.json.  A recorded
# measurement (a process cannot collect PMC counters on itself); reported only for the workload it was taken on.
PMC_TRAFFIC = {('bf16', 1024): {'bytes_per_launch': 26392558656.0, 'fetch_bytes': 26173237760.0, 'write_bytes': 219320896.0,
                                'algorithmic_bytes': 7.6e6,
                                'note': 'L2-to-fabric requests; Infinity-Cache hits are counted: the 5.9 MB bf16 weight '
                                        'stream outgrows a 4 MB per-XCD L2 and is re-fetched from the 256 MB Infinity '
                                        'Cache; 0.24 TB/s = 2.9 % of HBM peak, the launch is MFMA-bound',
                                'source': 'profiles/r1_bf16_v8_pmc_traffic.json'}}


def cpu_baseline(seconds_budget=15.0):
    """The CPU oracle (a restatement pinned bit-for-bit to the reference, oracle/make_golden.py)
    on the host cores: same 128-step schedule, a bounded batch."""
    from drakes_amd.det_init import det_state_dict
    from oracle import drakes_oracle as O
    from oracle import philox
    sd = det_state_dict(0)
    uni = lambda i, shp: torch.from_numpy(philox.step_uniforms(1, i, shp[0], shp[1]))
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    O.sample(sd, 2, L, 4, uni)                      # warm-up / first-touch
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.sample(sd, 16, L, NUM_STEPS, uni)             # sizing run: how many sequences fit the budget
    probe = time.perf_counter() - t0
    B = int(max(16, min(512, 16 * seconds_budget / max(probe, 1e-3)))) // 16 * 16
    t0 = time.perf_counter()
    O.sample(sd, B, L, NUM_STEPS, uni)
    dt = time.perf_counter() - t0
    return {'value': B / dt, 'unit': 'sequences/sec', 'cores': cores, 'kind': 'port',
            'sample': f'oracle.sample, batch {B}, 128 steps + noise removal, L=200, fp32 torch CPU, '
                      f'{cores} threads, {dt:.1f} s (warm-up {warm:.1f} s)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--batch', type=int, default=1024, help='sequences per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    local_rank = local_rank % max(torch.cuda.device_count(), 1)   # only differs in the one-GPU rehearsal
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    backend = os.environ.get('DRK_BENCH_BACKEND', 'nccl')   # 'gloo': rehearsal of the N>1 path on a one-GPU box
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)        # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    torch.manual_seed(0)
    model = drakes_amd.Diffusion(drakes_amd.default_config(length=L, steps=NUM_STEPS)).to(dev).eval()
    model.hip_dtype = args.dtype
    B = args.batch
    seed = 1234

    def one_step(k):
        return model._sample(eval_sp_size=B, seed=seed + k, sample_offset=rank * B)

    for k in range(args.warmup):
        one_step(k)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ops.PROFILE.enable(every=8 if args.dtype == 'fp32' else 1)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        toks = one_step(args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ops.PROFILE.disable()
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert toks.shape == (B, L) and int(toks.max()) <= 3

    if rank == 0:
        total_seqs = world * B * args.steps
        name, n_timed, avg_ms, flop_per_launch = prof
        achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms else None
        out = {
            'metric': 'sequences/sec, 128-step sampling of 200bp DNA, batch 1024, at 1/2/4/8 MI355X',
            'value': total_seqs / elapsed, 'unit': 'sequences/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.dtype == 'bf16' else 'f32',
            'data': 'synthetic (all-mask prior, random-init weights, in-kernel Philox noise)',
            'config': {'workload': 'configs[1]: Diffusion._sample, CNN denoiser (reference has no DiT), '
                                   '128 steps + noise removal, L=200, batch 1024 per GPU',
                       'global_batch': world * B, 'seq_len': L, 'parallelism': f'dp{world} (batch shards, no collective)'},
            'roofline': {'bound': 'mfma', 'kernel': name, 'achieved': achieved, 'peak': PEAK[args.dtype],
                         'unit': 'TFLOP/s', 'frac': (achieved / PEAK[args.dtype]) if achieved else None,
                         'traffic': (PMC_TRAFFIC.get((args.dtype, B)) or {}).get('bytes_per_launch'),
                         'traffic_detail': PMC_TRAFFIC.get((args.dtype, B)),
                         'launches_timed': n_timed, 'avg_launch_ms': avg_ms, 'flop_per_launch': flop_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0, N=1 only
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
