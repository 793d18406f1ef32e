import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import drakes_amd
from drakes_amd import fastpath
from drakes_amd.det_init import det_state_dict
dev = torch.device('cuda:0')
B, L = 4, 200
gen = np.random.Generator(np.random.PCG64(300))
x = torch.from_numpy(gen.integers(0, 5, (B, L))); x[0] = 4
out = {'x': x.numpy()}
sd = det_state_dict(7)
m = drakes_amd.Diffusion(drakes_amd.default_config()).to(dev); m.load_state_dict(sd); m.eval()
for layer in (-1, 0, 1, 4, 19):
    logits, _, feat = fastpath.backbone_logits_bf16(m.backbone, x.to(dev), dbg_layer=layer)
    out[f'feat_{layer}'] = feat.cpu().numpy()
out['logits'] = logits.cpu().numpy()
np.savez('gpurun_out/dbg_bf16.npz', **out)
