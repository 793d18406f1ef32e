"""The flat parameter-gradient path of _sample_finetune_gradient against the per-parameter one: same gradients, time.

    python tools/check_flat_grads.py
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import drakes_amd  # noqa: E402
dev = torch.device('cuda:0')
def run(flat, dtype):
    torch.manual_seed(0)
    m = drakes_amd.Diffusion(drakes_amd.default_config(steps=16)).to(dev)
    m.config.finetuning.truncate_steps = 6
    m.hip_soft_dtype = dtype
    m.hip_flat_grads = flat
    w = torch.randn(4, 200, device=dev)
    ts = []
    for rep in range(3):
        for p in m.parameters(): p.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m._sample_finetune_gradient(eval_sp_size=16, copy_flag_temp=None, seed=5)
        (out[0] * w.t()[None]).sum().backward()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, min(ts)
for dtype in ('fp32', 'bf16', 'bf16x3'):
    a, ta = run(False, dtype); b, tb = run(True, dtype)
    assert a.keys() == b.keys(), (set(a) ^ set(b))
    worst = max((a[k] - b[k]).abs().max().item() for k in a)
    scale = max(a[k].abs().max().item() for k in a)
    print(dtype, 'params with grad', len(a), 'max abs diff', worst, 'of', scale, 'time', ta, tb, flush=True)
