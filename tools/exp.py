"""Experiment harness for the fused kernel (not part of the product or the tests).

    python tools/exp.py build NAME [-DFLAG ...]      # CPU: builds drakes_amd/libexp_NAME.so
    DRK_LIB=... python tools/exp.py time [B] [steps] # GPU: times the fused kernel, prints ms and TF/s
    DRK_LIB=... python tools/exp.py stamps           # GPU: phase shares from a -DDRK_STAMPS build
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# DRK_PKG_ROOT=_ab_old: import the snapshot of an older package (its own library and packing) for A/B timing in one call
sys.path.insert(0, os.path.join(ROOT, os.environ['DRK_PKG_ROOT']) if os.environ.get('DRK_PKG_ROOT') else ROOT)

def build(name, flags):
    from drakes_amd import build as b
    objs = []
    for f, extra in b.SOURCES.items():
        obj = f'/tmp/exp_{name}_{f}.o'
        subprocess.check_call([b._hipcc()] + b.COMMON + extra + flags + ['-c', os.path.join(b.CSRC, f), '-o', obj])
        objs.append(obj)
    out = os.path.join(ROOT, 'drakes_amd', f'libexp_{name}.so')
    subprocess.check_call([b._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    print(out)

def setup(B, steps):
    import torch, drakes_amd
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = drakes_amd.Diffusion(drakes_amd.default_config(steps=steps)).to(dev).eval()
    m.hip_dtype = 'bf16'
    return m, dev

def time_it(B=1024, steps=16, reps=3):
    import torch
    m, dev = setup(B, steps)
    m._sample(eval_sp_size=B, seed=1)
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        t0 = time.perf_counter(); m._sample(eval_sp_size=B, seed=2 + r); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    flop = 2.0 * B * 200 * (45 * 128 + 20 * 9 * 128 * 128 + 128 * 128 + 128 * 5) * (steps + 1)
    print(f'LIB={os.environ.get("DRK_LIB","default")} B={B} steps={steps}: {t*1e3:.2f} ms  {flop/t/1e12:.1f} TF/s  ({t/(steps+1)*1e6/ (B/256):.1f} us per forward per sequence-slot)')

def stamps(B=1024, steps=8):
    import torch, numpy as np
    from drakes_amd import fastpath, ops, noise_schedule
    m, dev = setup(B, steps)
    bb = m.backbone
    table, sigma_last, _, _ = noise_schedule.step_table(m.noise, steps)
    tbias = bb.time_bias(torch.zeros(1, device=dev))
    mc = table[:, 2:4].contiguous().to(dev)
    out = torch.empty(B, 200, dtype=torch.int64, device=dev)
    dbg = torch.zeros(B * 200 * 128, dtype=torch.float32, device=dev)
    fastpath._launch(bb, tbias, 0, mc, None, None, out, None, -1, B, 200, steps, True, 5, 0, dbg, -2)
    torch.cuda.synchronize()
    raw = dbg.view(torch.int64)
    st = raw[: B * 8 * 4].reshape(B, 8, 4).cpu().numpy().astype(np.float64)
    ticks = raw[B * 32: B * 32 + B].cpu().numpy().astype(np.float64)          # 100 MHz
    p1 = raw[B * 33: B * 33 + B * 8].reshape(B, 8).cpu().numpy().astype(np.float64)
    p1c = raw[B * 41: B * 41 + B * 8].reshape(B, 8).cpu().numpy().astype(np.float64)
    head, step, stem = (raw[B * o: B * o + B * 8].reshape(B, 8).cpu().numpy().astype(np.float64) for o in (49, 57, 65))
    print('per forward, cycles by wave: head', (head.mean(0) / (steps + 1)).round(0), ' reverse step', (step.mean(0) / (steps + 1)).round(0),
          ' time-bias copy + stem MFMAs', (stem.mean(0) / (steps + 1)).round(0), ' stem epilogue', (st[:, :, 3].mean(0) / (steps + 1)).round(0))
    st[:, :, 3] += head + step + stem
    cyc = st[:, 0, 1].copy()                                                   # shader cycles of wave 0 over the same span
    print(f'in-kernel clock: {np.median(cyc / ticks) * 0.1:.3f} GHz (median over workgroups; min {np.min(cyc / ticks) * 0.1:.3f}, max {np.max(cyc / ticks) * 0.1:.3f})')
    st[:, :, 1] = p1
    st[:, :, 3] += p1c
    tot = st.sum(-1, keepdims=True)
    sh = st / tot
    names = ['mma', 'barrier 1 wait', 'epilogue pass 2 + barrier 2', 'stem/head/step (+ pass 1, listed below)']
    for w in range(8):
        print(f'wave {w}: total {tot[:, w].mean():.3e} cyc  ' + '  '.join(f'{n} {sh[:, w, i].mean()*100:.1f}%' for i, n in enumerate(names)))
    print('epilogue pass 1 (compute, up to barrier 1), cycles per epilogue, by wave:', (p1c.mean(0) / ((steps + 1) * 21)).round(0))
    print('wait at barrier 1, cycles per epilogue, by wave:', (p1.mean(0) / ((steps + 1) * 21)).round(0))
    per_fwd_layer = st.mean(0) / ((steps + 1) * 20)
    print('cycles per layer (avg over waves):', per_fwd_layer.mean(0)[:3], ' per forward other:', st.mean(0).mean(0)[3] / (steps + 1))

if __name__ == '__main__':
    cmd = sys.argv[1]
    if cmd == 'build':
        build(sys.argv[2], sys.argv[3:])
    elif cmd == 'time':
        time_it(*(int(x) for x in sys.argv[2:]))
    elif cmd == 'stamps':
        stamps()
