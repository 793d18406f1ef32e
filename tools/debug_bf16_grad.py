"""Per-parameter gradient error of the bf16 differentiable denoiser against oracle autograd (diagnostic)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import drakes_amd
from drakes_amd.det_init import det_state_dict
from oracle import drakes_oracle as O
dev = torch.device('cuda:0')
sd = det_state_dict(7)
B, L = 4, 200
gen = np.random.Generator(np.random.PCG64(40))
x = torch.from_numpy(gen.integers(0, 5, (B, L)))
Gm = torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32))
sigma = torch.zeros(B)
sdg = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in sd.items()}
(O.cnn_forward(sdg, x, sigma) * Gm).sum().backward()
from drakes_amd import autograd as AG
from oracle import bf16_emulation as E
AG._BackboneBf16Fn.keep_gates = True
for dt in ('bf16',):
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=L)).to(dev)
    m.load_state_dict(sd)
    m.backbone.train_dtype = dt
    (m.backbone(x.to(dev), sigma.to(dev)) * Gm.to(dev)).sum().backward()
    print(dt)
    for name, p in m.named_parameters():
        r = sdg[name].grad
        if r is None or 'time_layers' in name:
            continue
        e = (p.grad.cpu() - r).abs().max().item() / r.abs().max().item()
        rms = ((p.grad.cpu() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
        print(f'  {name:40s} max-rel {e:.4f}  rms-rel {rms:.4f}')

masks, y = AG._BackboneBf16Fn.last_gates
masks = [mk.cpu().float() for mk in masks]
hm = (y > 0).cpu().float()
for label, kw in (('emulation, own gates', {}), ('emulation, kernel gates', dict(masks=masks, head_mask=hm))):
    sdg = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in sd.items()}
    (E.cnn_forward_train_bf16(sdg, x, sigma, **kw) * Gm).sum().backward()
    print(label)
    for name, p in m.named_parameters():
        r = sdg[name].grad
        if r is None or 'time_layers' in name or 'norms' in name:
            continue
        e = (p.grad.cpu() - r).abs().max().item() / r.abs().max().item()
        rms = ((p.grad.cpu() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
        print(f'  {name:40s} max-rel {e:.4f}  rms-rel {rms:.4f}')
