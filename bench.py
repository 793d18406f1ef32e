"""bench.py — throughput of the reverse-sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--dtype bf16|fp32] [--batch 1024] [--workload sample|long|finetune]

One "step" = one full `Diffusion._sample` call (128 reverse steps + the noise-removal forward
= 129 denoiser forwards) over a batch of 1024 200-bp sequences per GPU: BASELINE.json configs[1].
Weights are random-init (default torch init of the reference architecture), data is synthetic
(the sampler starts from the all-mask state), RNG is the in-kernel Philox stream.

N > 1: one rank per GPU, each sampling its own shard of the global batch (rows keyed by GLOBAL sample
index, no data-path collective) -> weak scaling.  Two ways in, same code after the rendezvous:
  * under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment);
  * plain `python bench.py --gpus N`: this process starts N children itself (one per LOCAL_RANK, rendezvous on
    127.0.0.1) BEFORE it makes any GPU call, waits, relays rank 0's JSON line and exits non-zero if any child does.
The collective backend is 'nccl' (= RCCL over xGMI).  On a box with fewer GPUs than ranks (the one-GPU rehearsal)
the ranks share a device and the group falls back to 'gloo', because two RCCL ranks cannot share a GPU; the line
says so in `collective`.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the timed region),
`cpu_baseline` (the CPU oracle on a bounded sample of the same workload, N=1 only) and `collective`
(`ranks_seen` = an all_reduce of ones over the group: proof the collective library saw every rank).

Other workloads (not the driver's headline; same JSON shape, their own metric names):
  --workload long      BASELINE.json configs[4] on one GPU: L=1024, 512 steps, hipGraph-free layer-per-launch bf16 kernels
  --workload finetune  BASELINE.json configs[2]: `_sample_finetune_gradient` (K=50) + backward, batch 128
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L = 200
NUM_STEPS = 128
# Dense FLOP count as the reference computes it
FLOP_PER_POS_LAYER = 2 * 128 * 128 * 9            # 294 912
FLOP_PER_POS_FWD = 11520 + 20 * FLOP_PER_POS_LAYER + 34048   # 5.944 MFLOP
PEAK = {'bf16': 2500.0, 'fp32': 157.3}            # TFLOP/s dense MFMA, MI355X_MICROARCH.md
# HBM-side bytes of ONE k_sample_bf16 launch at exactly this workload (B=1024, L=200, 129 forwards), from rocprofv3 PMC
# passes (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes
# for 16-B-per-lane loads on gfx950): tools/pmc_traffic.sh.  A recorded measurement (a process cannot collect PMC
# counters on itself): it names the library version it was taken on and is printed ONLY when drk_version() still is
# that version — a changed kernel reports traffic null until the passes are re-run.
PMC_TRAFFIC = {('bf16', 1024): {'drk_version': 311,
                                'bytes_per_launch': 27219465664.0, 'fetch_bytes': 26716473088.0, 'write_bytes': 502992576.0,
                                'algorithmic_bytes': 7.6e6,
                                'note': 'L2-to-fabric requests; Infinity-Cache hits are counted: the 5.9 MB bf16 weight '
                                        'stream outgrows a 4 MB per-XCD L2 and is re-fetched from the 256 MB Infinity '
                                        'Cache; 0.25 TB/s = 3.1 % of HBM peak, the launch is MFMA-bound; the writes are '
                                        'scratch spills of the per-forward stem/head block (140 B/lane reserved; 2 dwords per lane and forward '
                                        'actually written), tokens out are 1.6 MB',
                                'source': 'profiles/r3_bf16_v311_pmc_traffic.json'}}


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def cpu_baseline():
    """The CPU oracle (a restatement pinned bit-for-bit to the reference, oracle/make_golden.py) on the host cores,
    as BASELINE.md section 3 prescribes: configs[0] = batch 64, 128 steps + noise removal, L=200, all the cores this
    process may use, 1 warm-up, median of 3 full calls."""
    from drakes_amd.det_init import det_state_dict
    from oracle import drakes_oracle as O
    from oracle import philox
    sd = det_state_dict(0)
    uni = lambda i, shp: torch.from_numpy(philox.step_uniforms(1, i, shp[0], shp[1]))
    cores = torch.get_num_threads()
    B = 64
    t0 = time.perf_counter()
    O.sample(sd, B, L, 8, uni)                      # warm-up at the timed shape (thread pool, first touch), 8 steps
    warm = time.perf_counter() - t0
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.sample(sd, B, L, NUM_STEPS, uni)
        ts.append(time.perf_counter() - t0)
    dt = _median(ts)
    return {'value': B / dt, 'unit': 'sequences/sec', 'cores': cores, 'kind': 'port',
            'sample': f'oracle.sample (configs[0]), batch {B}, 128 steps + noise removal, L=200, fp32 torch CPU, '
                      f'{cores} threads; 1 warm-up ({warm:.1f} s, 8 steps), median of 3 calls: '
                      + ', '.join(f'{t:.2f}' for t in ts) + ' s'}


# ---------------------------------------------------------------------------------------------- self-launch
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """Starts one child per rank (what torch.distributed.run would do), relays rank 0's stdout, returns the exit code.
    Runs before this process has touched the GPU: the parent never initialises HIP.  Every child is polled: the first
    non-zero exit, or the wall-clock limit (DRK_BENCH_TIMEOUT seconds, default 1500), ends the others (our own children,
    by PID) instead of leaving rank 0 in a rendezvous until the process-group timeout."""
    import tempfile
    port = _free_port()
    limit = float(os.environ.get('DRK_BENCH_TIMEOUT', '1500'))
    procs = []
    rc = 0
    with tempfile.TemporaryFile() as out0, tempfile.TemporaryFile() as err_rest:
        try:
            for r in range(n):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                           MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONUNBUFFERED='1')
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                              stdout=out0 if r == 0 else err_rest))
            t0 = time.monotonic()
            while True:
                codes = [p.poll() for p in procs]
                bad = [c for c in codes if c not in (None, 0)]
                if bad:
                    rc = bad[0]
                    break
                if all(c == 0 for c in codes):
                    break
                if time.monotonic() - t0 > limit:
                    rc = 124
                    print(f'bench.py: ranks still running after {limit:.0f} s', file=sys.stderr)
                    break
                time.sleep(0.1)
        finally:
            for p in procs:                               # our own children, by PID
                if p.poll() is None:
                    p.kill()
                    rc = rc or 1
            for p in procs:
                p.wait()
        out0.seek(0)
        sys.stdout.write(out0.read().decode(errors='replace'))
        sys.stdout.flush()
        if rc:
            err_rest.seek(0)
            sys.stderr.write(err_rest.read().decode(errors='replace')[-2000:])
    if rc:
        print(f'bench.py: a rank exited with {rc}', file=sys.stderr)
    return rc


class _stdout_to_stderr:
    """RCCL prints a version banner on the C stdout when its first communicator comes up; the contract is ONE JSON line
    on stdout, so file descriptor 1 points at stderr while a process group is being created and first used."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)          # C stdio buffers of the libraries that printed
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)


def _rccl_world1_probe(dev):
    """N=1: join an 'nccl' (RCCL) group of one rank and push (a) a one, (b) a gradient-bucket-sized fp32 tensor through
    all_reduce, so that the library the N>1 runs depend on is loaded and exercised on this box too."""
    import torch.distributed as dist
    info = {'backend': 'nccl', 'world': 1}
    try:
        with _stdout_to_stderr():
            dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1,
                                    device_id=dev)
            one = torch.ones(1, device=dev)
            dist.all_reduce(one)
            info['ranks_seen'] = int(one.item())
            flat = torch.ones(3326597, dtype=torch.float32, device=dev)     # the fine-tune gradient bucket (13.3 MB)
            dist.all_reduce(flat)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                dist.all_reduce(flat)
            torch.cuda.synchronize()
            info['allreduce_13MB_us'] = (time.perf_counter() - t0) / 5 * 1e6
            info['ok'] = bool(flat[0].item() == 1.0)
            dist.destroy_process_group()
    except Exception as e:                              # a box without a usable RCCL must not lose the bench line
        info['ok'] = False
        info['error'] = f'{type(e).__name__}: {e}'[:300]
    return info


# ---------------------------------------------------------------------------------------------- workloads
def workload_sample(args, model, rank, world, barrier):
    import drakes_amd
    from drakes_amd import fastpath, ops
    B = args.batch
    seed = 1234

    def one_step(k):
        return model._sample(eval_sp_size=B, seed=seed + k, sample_offset=rank * B)

    for k in range(args.warmup):
        one_step(k)
    ops.PROFILE.enable(every=8 if args.dtype == 'fp32' else 1)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        toks = one_step(args.warmup + k)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ops.PROFILE.disable()
    assert toks.shape == (B, L) and int(toks.max()) <= 3 and int(toks.min()) >= 0
    name, n_timed, avg_ms, flop_per_launch = prof
    achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms else None
    roof = {'bound': 'mfma', 'kernel': name, 'achieved': achieved, 'peak': PEAK[args.dtype], 'unit': 'TFLOP/s',
            'frac': (achieved / PEAK[args.dtype]) if achieved else None, 'traffic': None,
            'launches_timed': n_timed, 'avg_launch_ms': avg_ms, 'flop_per_launch': flop_per_launch,
            'numerator': 'dense algorithmic FLOPs, as the reference computes them (BASELINE.md section 4)'}
    if args.dtype == 'bf16':
        # what the kernel's MFMAs actually execute: dead (tile, tap) pairs are skipped, L=200 is padded to 208
        ex = fastpath.executed_flop_fraction(model.backbone, L)
        roof['executed_flop_frac'] = ex
        roof['frac_executed'] = roof['frac'] * ex if roof['frac'] else None
    rec = PMC_TRAFFIC.get((args.dtype, B))
    if rec is not None:
        from drakes_amd import _lib
        ver = _lib.load().drk_version()
        if rec['drk_version'] == ver:
            roof['traffic'] = rec['bytes_per_launch']
            roof['traffic_detail'] = rec
        else:
            roof['traffic_detail'] = {'stale': f"recorded on library version {rec['drk_version']}, this is {ver}: "
                                               're-run tools/pmc_traffic.sh'}
    out = {'metric': 'sequences/sec, 128-step sampling of 200bp DNA, batch 1024, at 1/2/4/8 MI355X',
           'value': world * B * args.steps / elapsed, 'unit': 'sequences/sec',
           'dtype': 'bf16' if args.dtype == 'bf16' else 'f32',
           'data': 'synthetic (all-mask prior, random-init weights, in-kernel Philox noise)',
           'config': {'workload': 'configs[1]: Diffusion._sample, CNN denoiser (reference has no DiT), '
                                  '128 steps + noise removal, L=200, batch 1024 per GPU',
                      'global_batch': world * B, 'seq_len': L, 'parallelism': f'dp{world} (batch shards, no collective)'},
           'roofline': roof}
    return out, elapsed


def secondary_fp32(model, B=1024, dtype='fp32'):
    """The token-exact fp32 path (or 'bf16x3': the same loop with fp32-grade convolutions on the bf16 matrix cores) on the
    same workload, in the same run: one warm-up call, one timed call."""
    model.hip_dtype = dtype
    try:
        model._sample(eval_sp_size=B, seed=1)         # warm-up at the timed shape: work buffers exist before the clock starts
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model._sample(eval_sp_size=B, seed=2)
        torch.cuda.synchronize()
        return B / (time.perf_counter() - t0)
    finally:
        model.hip_dtype = 'bf16'


# The dominant launch of configs[4] (k_group_bf16<8,true>: layers 8..19 + head, B=256, L=1024): tools/pmc_group.sh,
# profiles/r3_group_v311_pmc.json.  Printed only while drk_version() is the version it was taken on.
PMC_TRAFFIC_LONG = {'drk_version': 311, 'bytes_per_launch': 386126464.0, 'fetch_bytes': 303667968.0, 'write_bytes': 82458496.0,
                    'launch': 'the largest of the three: layers 8..19 + head, stride 4 (the two halo launches: 360 MB each)',
                    'algorithmic_bytes': 256 * 1024 * (512 + 20.0),
                    'note': 'L2-to-fabric requests (FETCH_SIZE doubled per MI355X_MICROARCH.md); beyond the 134 MB of residual '
                            'stream: the 3.5 MB weight stream of the 12 layers re-fetched from the Infinity Cache and 49 '
                            'spilled registers (76 MB of scratch writes); 0.5 TB/s, the launch is MFMA-bound'}


def _time_layer_kernel(model, B, Ll, reps=20):
    import ctypes
    from drakes_amd import _lib, fastpath, ops
    bb = model.backbone
    st = fastpath._packed(bb)
    dev = next(bb.parameters()).device
    feat = torch.randn(B, Ll, 128, device=dev)
    h0 = torch.randn(B, Ll, 128, device=dev).to(torch.bfloat16)
    h1 = torch.empty_like(h0)
    tbias = bb.time_bias(torch.zeros(1, device=dev))
    vec = st['vecs']
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    a = (P(h0), P(feat), P(feat), P(h1), P(st['wstream']), P(vec[3]), P(tbias[0, 1]), 0, P(vec[4]), P(vec[5]), B, Ll, 9, 1, 1, 1,
         None, None, None, ops._stream())
    for _ in range(3):
        _lib.call('drk_layer_bf16', *a)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s_, e_ in evs:
        s_.record()
        _lib.call('drk_layer_bf16', *a)
        e_.record()
    torch.cuda.synchronize()
    ms = sum(s_.elapsed_time(e_) for s_, e_ in evs) / reps
    return 'k_layer_bf16 (one residual layer, timed alone after the run)', reps, ms, 2.0 * B * Ll * 9 * 128 * 128


def _time_group_kernels(model, B, Ll, reps=10):
    """The launches of the grouped long-sequence forward (drk_layer_group_bf16; the first with the stem, the last with the
    head) timed after the run, same shapes, HIP events on their stream: (a) `reps` forwards captured into one hipGraph
    and replayed, as the timed region runs them -> ms per forward; (b) every launch on its own, back to back (they run
    5-10 % slower that way than inside the captured sequence).  Returns (ms per forward, [(plan entry, ms alone, dense
    FLOP)]), or None when the shape takes the layer path."""
    from drakes_amd import _lib, fastpath
    bb = model.backbone
    plan = fastpath.group_plan(bb, Ll)
    if plan is None:
        return None
    dev = next(bb.parameters()).device
    x = torch.randint(0, 5, (B, Ll), device=dev)
    work = {}
    tbias = bb.time_bias(torch.zeros(1, device=dev))
    fastpath.forward_bf16_groups(bb, x, tbias, work, plan)            # fills the buffers with a real forward's values
    launches, _, _ = fastpath.group_launches(bb, x, tbias, work, plan)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        captured, _, _ = fastpath.group_launches(bb, x, tbias, work, plan)     # (the argument tuples carry the stream)
        for _ in range(reps):
            for _, a in captured:
                _lib.call('drk_layer_group_bf16', *a)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(3):
        graph.replay()
    e_.record()
    torch.cuda.synchronize()
    fwd_ms = s_.elapsed_time(e_) / (3 * reps)
    out = []
    for entry, a in launches:
        for _ in range(2):
            _lib.call('drk_layer_group_bf16', *a)
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(reps):
            _lib.call('drk_layer_group_bf16', *a)
        e_.record()
        torch.cuda.synchronize()
        out.append((entry, s_.elapsed_time(e_) / reps, 2.0 * B * Ll * 9 * 128 * 128 * entry[1]))
    return fwd_ms, out


def workload_long(args, rank, world, dev, barrier):
    """configs[4] per GPU: L=1024, 512 steps, bf16 kernels for sequences beyond the fused loop's 224 positions: the
    residual layers several per launch (drk_layer_group_bf16)."""
    import drakes_amd
    from drakes_amd import ops
    Ll, steps, B = 1024, 512, args.batch if args.batch != 1024 else 256
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=Ll, steps=steps)).to(dev).eval()
    m.hip_dtype = args.dtype
    m.hip_graph = True
    for k in range(args.warmup):
        m._sample(eval_sp_size=B, seed=k, sample_offset=rank * B)
    ops.PROFILE.enable(every=97)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        m._sample(eval_sp_size=B, seed=10 + k, sample_offset=rank * B)
    barrier()
    elapsed = time.perf_counter() - t0
    name, n, ms, fl = ops.PROFILE.disable()
    rows = B * Ll
    roof = {'bound': 'mfma', 'kernel': name, 'launches_timed': n, 'avg_launch_ms': ms}
    groups = _time_group_kernels(m, B, Ll) if (not n and args.dtype == 'bf16') else None
    if groups:
        # the captured step replays without passing the launch brackets.  Dominant kernel: k_group_bf16, whose three
        # launches are the forward (two halo launches of four dilation-1 layers, one stride-4 launch of the twelve
        # dilation-4/16/64 layers and the head); MFMA-bound: a forward moves 2.1 KB per row through HBM for 20 x 295 kFLOP
        fwd_ms, per = groups
        flop = sum(f for _, _, f in per)
        g, gms, gfl = max(per, key=lambda t: t[1])
        roof = {'bound': 'mfma', 'kernel': f'k_group_bf16 (the {len(per)} launches of one forward, replayed from a hipGraph after the run)',
                'launches_timed': 30 * len(per), 'avg_launch_ms': fwd_ms / len(per), 'forward_ms': fwd_ms,
                'achieved': flop / fwd_ms / 1e9, 'peak': 2500.0, 'unit': 'TFLOP/s',
                'traffic': None, 'flop_per_row_and_layer': 2 * 9 * 128 * 128, 'flop_per_forward': flop,
                'hbm_GBps': rows * 2068 / (fwd_ms * 1e-3) / 1e9,
                'launches_alone_ms': [[list(e), round(t, 4)] for e, t, _ in per],
                'largest_launch_alone': {'plan': list(g), 'ms': gms, 'TFLOPs_dense': gfl / gms / 1e9}}
        roof['frac'] = roof['achieved'] / roof['peak']
        from drakes_amd import _lib
        if B == 256 and PMC_TRAFFIC_LONG['drk_version'] == _lib.load().drk_version():
            roof['traffic'] = PMC_TRAFFIC_LONG['bytes_per_launch']
            roof['traffic_detail'] = PMC_TRAFFIC_LONG
    elif not n and args.dtype == 'bf16':
        name, n, ms, fl = _time_layer_kernel(m, B, Ll)
        roof = {'bound': 'hbm', 'kernel': name, 'launches_timed': n, 'avg_launch_ms': ms}
    if 'frac' not in roof and ms and name and 'layer' in name:
        per_row = getattr(ops, 'LAYER_BYTES_PER_ROW', 1536)
        roof.update({'achieved': rows * per_row / (ms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'traffic': None,
                     'bytes_per_row': per_row, 'TFLOPs_dense': fl / ms / 1e9})
        roof['frac'] = roof['achieved'] / roof['peak']
    out = {'metric': 'sequences/sec, 512-step sampling of 1024bp sequences (configs[4], per-GPU replica)',
           'value': world * B * args.steps / elapsed, 'unit': 'sequences/sec', 'dtype': 'bf16' if args.dtype == 'bf16' else 'f32',
           'data': 'synthetic (all-mask prior, random-init weights, in-kernel Philox noise)',
           'config': {'workload': f'configs[4]: Diffusion._sample, L={Ll}, {steps} steps + noise removal, batch {B} per GPU, '
                                  'captured step (hipGraph)', 'global_batch': world * B, 'seq_len': Ll,
                      'parallelism': f'dp{world} (replicas, no collective)'},
           'roofline': roof}
    return out, elapsed


def workload_finetune(args, rank, world, dev, barrier, backend):
    """configs[2] per GPU and configs[3] across GPUs.  One step = one data-parallel fine-tune micro-step of the reference's
    loop (finetune_reward_bp.py:40-43, 83-93) with a synthetic linear reward: `_sample_finetune_gradient` (78 hard + 50
    soft steps) on this rank's batch shard, backward, ONE all-reduce of the flat gradient bucket (drakes_amd.dp.GradBucket,
    13.3 MB) -> mean -> clip_grad_norm_ -> Adam.step.  At N = 1 the all-reduce still runs, through an RCCL group of one."""
    import torch.distributed as dist
    import drakes_amd
    from drakes_amd import dp
    B = args.batch if args.batch != 1024 else 128
    cfg = drakes_amd.default_config(length=L, steps=NUM_STEPS, truncate_steps=50)
    m = drakes_amd.Diffusion(cfg).to(dev).train()
    m.hip_dtype = args.dtype
    rw = torch.randn(B, 4, L, device=dev)
    optim = torch.optim.Adam(m.parameters(), lr=1e-4)
    bucket = dp.GradBucket(m)
    timed = backend == 'nccl' and dist.is_initialized()
    ar_events, ar_host = [], []

    def one(k, record):
        bucket.zero_()
        out = m._sample_finetune_gradient(eval_sp_size=B, seed=100 + k, sample_offset=rank * B)[0]
        (-(out.permute(0, 2, 1) * rw).sum((1, 2)).mean()).backward()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.allreduce_mean_(force=True)
            e1.record()
            if record:
                ar_events.append((e0, e1))
        else:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bucket.allreduce_mean_(force=True)
            torch.cuda.synchronize()
            if record:
                ar_host.append(time.perf_counter() - t0)
        bucket.clip_grad_norm_(1.0)
        optim.step()

    for k in range(args.warmup):
        one(k, False)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one(args.warmup + k, True)
    barrier()
    elapsed = time.perf_counter() - t0
    ar_us = [a.elapsed_time(b) * 1e3 for a, b in ar_events] if timed else [t * 1e6 for t in ar_host]
    # dense FLOPs: 128 forwards + 50 x (input-gradient + weight-gradient convs) = 128 + 100 forward-equivalents
    flop = B * L * FLOP_PER_POS_FWD * (NUM_STEPS + 2 * 50)
    per = elapsed / args.steps
    from drakes_amd import _lib
    cfgname = 'configs[2]' if world == 1 else 'configs[3]'
    out = {'metric': 'fine-tune 128-sequence shard passes/sec, all GPUs (configs[2] per GPU, configs[3] across GPUs: '
                     '_sample_finetune_gradient K=50 + backward + gradient all-reduce + clip + Adam per pass)',
           'value': world * args.steps / elapsed, 'unit': 'passes/sec', 'dtype': 'bf16' if args.dtype == 'bf16' else 'f32',
           'data': 'synthetic (random-init weights, linear reward, in-kernel Philox noise)',
           'config': {'workload': f'{cfgname}: batch {B} per GPU, L=200, 128 steps, K=50, one optimizer step per pass',
                      'global_batch': world * B, 'seq_len': L,
                      'parallelism': f'dp{world} (batch shards; one {bucket.flat.numel() * 4 / 1e6:.1f} MB gradient all-reduce per step)'},
           'seconds_per_pass': per, 'sequences_per_sec': world * B / per, 'peak_mem_GiB': torch.cuda.max_memory_allocated() / 2 ** 30,
           'drk_version': _lib.load().drk_version(),
           'allreduce_in_loop': {'backend': backend if dist.is_initialized() else None, 'bytes': bucket.flat.numel() * 4,
                                 'us_median': _median(ar_us) if ar_us else None, 'us_all': [round(u, 1) for u in ar_us],
                                 'timed_by': 'HIP events on the stream' if timed else 'host clock around a synchronised call '
                                             '(gloo staged through the host: ranks share a GPU)'},
           'roofline': {'bound': 'mfma', 'kernel': 'whole pass', 'achieved': flop / per / 1e12, 'peak': PEAK[args.dtype],
                        'unit': 'TFLOP/s', 'frac': flop / per / 1e12 / PEAK[args.dtype], 'traffic': None}}
    return out, elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--batch', type=int, default=1024, help='sequences per GPU')
    ap.add_argument('--workload', default='sample', choices=['sample', 'long', 'finetune'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or os.environ.get('DRK_BENCH_SPAWN')):
        raise SystemExit(spawn_ranks(args.gpus))           # no GPU call has been made in this process

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    n_dev = max(torch.cuda.device_count(), 1)               # device_count() does not initialise the GPU
    shared_device = world > n_dev                           # the one-GPU rehearsal: ranks share a card
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    backend = os.environ.get('DRK_BENCH_BACKEND') or ('gloo' if shared_device else 'nccl')
    collective = {'backend': backend, 'world': world}
    import torch.distributed as dist
    if world > 1:
        with _stdout_to_stderr():
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=dev)        # RCCL over xGMI
            else:
                dist.init_process_group(backend)
            ones = torch.ones(1, device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(ones)
            torch.cuda.synchronize()
        collective['ranks_seen'] = int(ones.item())
        if shared_device:
            collective['note'] = f'{world} ranks share {n_dev} GPU(s): rehearsal, RCCL needs one device per rank'
        assert collective['ranks_seen'] == world

    import drakes_amd
    torch.manual_seed(0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    model = None
    if args.workload == 'sample':
        model = drakes_amd.Diffusion(drakes_amd.default_config(length=L, steps=NUM_STEPS)).to(dev).eval()
        model.hip_dtype = args.dtype
        out, elapsed = workload_sample(args, model, rank, world, barrier)
    elif args.workload == 'long':
        out, elapsed = workload_long(args, rank, world, dev, barrier)
    else:
        if world == 1:                       # the step's all-reduce runs through an RCCL group of one
            try:
                with _stdout_to_stderr():
                    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1,
                                            device_id=dev)
                    one = torch.ones(1, device=dev)
                    dist.all_reduce(one)
                    torch.cuda.synchronize()
                collective['ranks_seen'] = int(one.item())
            except Exception as e:
                collective = {'backend': None, 'world': 1, 'error': f'{type(e).__name__}: {e}'[:300]}
        out, elapsed = workload_finetune(args, rank, world, dev, barrier, backend)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out['value'] *= elapsed / t.item()                  # every rank's work over the SLOWEST rank's time
        elapsed = t.item()

    if rank == 0:
        line = {'metric': out.pop('metric'), 'value': out.pop('value'), 'unit': out.pop('unit'), 'n_gpus': world,
                'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None}
        line.update(out)
        if 'drk_version' not in line:
            from drakes_amd import _lib
            line['drk_version'] = _lib.load().drk_version()
        if world == 1:
            if args.workload != 'finetune':
                collective = _rccl_world1_probe(dev)
            if args.workload == 'sample' and args.dtype == 'bf16' and not args.no_secondary:
                line['secondary'] = {'fp32_token_exact_seq_per_s': secondary_fp32(model),
                                     'bf16x3_fp32_grade_seq_per_s': secondary_fp32(model, dtype='bf16x3'),
                                     'note': 'same workload (B=1024, 128 steps + noise removal) on the exact-fp32 kernels, '
                                             'whose tokens equal the reference bit for bit, and with their convolutions '
                                             'at fp32 grade on the bf16 matrix cores (three bf16 planes per operand; tokens '
                                             'equal on the fixtures); 1 warm-up + 1 timed call each'}
            if not args.no_cpu_baseline and args.workload == 'sample':      # rank 0, N=1 only
                line['cpu_baseline'] = cpu_baseline()
        line['collective'] = collective
        print(json.dumps(line), flush=True)
    if world > 1 or dist.is_initialized():
        with _stdout_to_stderr():
            if world > 1:
                dist.barrier()
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
