"""One short launch of the fused sampler for PMC passes: python3 tools/pmc_run.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import drakes_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
torch.manual_seed(0)
m = drakes_amd.Diffusion(drakes_amd.default_config(steps=steps)).to('cuda:0').eval()
m.hip_dtype = 'bf16'
x = m._sample(eval_sp_size=B, seed=1)
torch.cuda.synchronize()
print('ok', x.shape, int(x.max()))
