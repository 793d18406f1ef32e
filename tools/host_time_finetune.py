"""Host (enqueue) time against device time of one configs[2] pass: how far ahead of the GPU the Python side runs.
python tools/host_time_finetune.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drakes_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
m = drakes_amd.Diffusion(drakes_amd.default_config(length=200, steps=128, truncate_steps=50)).to(dev).train()
m.hip_dtype = 'bf16'
rw = torch.randn(B, 4, 200, device=dev)
for r in range(5):
    m.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m._sample_finetune_gradient(eval_sp_size=B, seed=100 + r)[0]
    loss = -(out.permute(0, 2, 1) * rw).sum((1, 2)).mean()
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f'B={B} rep {r}: host forward {1e3 * (t1 - t0):.1f} ms, host backward {1e3 * (t2 - t1):.1f} ms, host total {1e3 * (t2 - t0):.1f} ms, '
          f'device done after {1e3 * (t3 - t0):.1f} ms', flush=True)
