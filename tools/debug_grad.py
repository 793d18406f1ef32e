import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import drakes_amd
from drakes_amd.det_init import det_state_dict
from oracle import drakes_oracle as O
dev = torch.device('cuda:0'); sd = det_state_dict(7)
gen = np.random.Generator(np.random.PCG64(5))
B, L = 3, 77
for tc, soft in ((True, True), (True, False), (False, True)):
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=L, time_conditioning=tc)).to(dev); m.load_state_dict(sd)
    x = torch.from_numpy(gen.dirichlet(np.ones(5) * 0.5, (B, L)).astype(np.float32)) if soft else torch.from_numpy(gen.integers(0, 5, (B, L)))
    sigma = torch.tensor([0.3, 1.7, 4.0]) if tc else torch.zeros(B)
    Gm = torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32))
    sdg = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in sd.items()}
    (O.cnn_forward(sdg, x, sigma) * Gm).sum().backward()
    for rep in range(2):
        m.zero_grad()
        logits = m.backbone(x.to(dev), sigma.to(dev))
        (logits * Gm.to(dev)).sum().backward()
        bad = []
        for name, p in m.named_parameters():
            ref = sdg[name].grad
            if ref is None: continue
            e = (p.grad.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            if e > 1e-3: bad.append((name, round(e, 4)))
        print('tc', tc, 'soft', soft, 'rep', rep, 'bad:', bad)
