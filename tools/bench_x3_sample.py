import sys, time, torch
sys.path.insert(0, '/root/repo')
import drakes_amd
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = drakes_amd.Diffusion(drakes_amd.default_config()).to(dev).eval()
for dt in ('fp32', 'bf16x3'):
    m.hip_dtype = dt
    m._sample(eval_sp_size=1024, seed=1, num_steps=8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = m._sample(eval_sp_size=1024, seed=2)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(dt, f'{t:.3f} s  {1024 / t:.0f} sequences/s', flush=True)
    if dt == 'fp32': ref = x
print('token mismatch bf16x3 vs fp32 at B=1024:', (x != ref).float().mean().item())
