"""Models narrower than the kernels' 128 lanes (the reference takes hidden_dim from args: models/dnaconv.py:46-63;
SURVEY 8c asks for a hidden_dim = 32 model for cheap tests).  The narrow model keeps the reference's parameter shapes
and runs zero-padded on the exact-fp32 kernels; it is checked against the oracle on the SAME narrow state dict: logits
(2e-4 absolute, 1e-4 relative, as tests/test_gpu_backbone.py), sampled tokens (equal), parameter and input gradients
(2e-3 of each tensor's max, as tests/test_gpu_finetune.py)."""
import numpy as np
import pytest
import torch

import drakes_amd
from oracle import drakes_oracle as O
from oracle import philox

pytestmark = pytest.mark.gpu
ATOL, RTOL, GTOL = 2e-4, 1e-4, 2e-3


def _model(dev, width, **kw):
    torch.manual_seed(width)
    m = drakes_amd.Diffusion(drakes_amd.default_config(hidden_dim=width, **kw))
    with torch.no_grad():                       # away from the all-default init: LayerNorm affine and biases that matter
        for n, p in m.named_parameters():
            if n.endswith('norms.0.weight') or 'norms' in n:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m.to(dev), sd


@pytest.mark.parametrize('width', [32, 64, 8])
def test_narrow_logits_vs_oracle(dev, width):
    m, sd = _model(dev, width)
    assert sd['backbone.convs.0.weight'].shape == (width, width, 9) and sd['backbone.time_embedder.0.W'].shape == (width // 2,)
    gen = np.random.Generator(np.random.PCG64(width))
    for B, L in ((3, 200), (2, 77)):
        x = torch.from_numpy(gen.integers(0, 5, (B, L)))
        soft = torch.from_numpy(gen.dirichlet(np.ones(5) * 0.5, (B, L)).astype(np.float32))
        sigma = torch.from_numpy(gen.uniform(0.01, 6.0, B).astype(np.float32))
        with torch.no_grad():
            for inp in (x, soft):
                for sg in (torch.zeros(B), sigma):
                    want = O.cnn_forward(sd, inp, sg).numpy()
                    got = m.backbone(inp.to(dev), sg.to(dev)).cpu().numpy()
                    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


def test_narrow_sample_vs_oracle(dev):
    width, L, steps, B, seed = 32, 200, 24, 6, 11
    m, sd = _model(dev, width, length=L, steps=steps)
    uni = lambda i, shp: torch.from_numpy(philox.step_uniforms(seed, i, shp[0], shp[1]))
    want = O.sample(sd, B, L, steps, uni)
    got = m._sample(eval_sp_size=B, seed=seed).cpu()
    rate = (got != want).float().mean().item()
    print('narrow-model token mismatch rate:', rate)
    assert rate == 0.0
    m.hip_graph = True
    assert torch.equal(m._sample(eval_sp_size=B, seed=seed).cpu(), got)


@pytest.mark.parametrize('soft', [False, True])
def test_narrow_backward_vs_oracle(dev, soft):
    width, B, L = 32, 2, 45
    m, sd = _model(dev, width, length=L, time_conditioning=True)
    sigma = torch.tensor([0.3, 4.0])
    bsd = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    for seed in range(400):                    # differentiate away from ReLU's kink (tests/test_gpu_finetune.py)
        gen = np.random.Generator(np.random.PCG64(seed))
        x = torch.from_numpy(gen.dirichlet(np.ones(5) * 0.5, (B, L)).astype(np.float32)) if soft else \
            torch.from_numpy(gen.integers(0, 5, (B, L)))
        pre = []
        with torch.no_grad():
            O.cnn_forward(sd, x, sigma, preacts=pre)
        if min(float(c.abs().min()) for c in pre) > 5e-6:
            break
    Gm = torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32))
    sdg = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in sd.items()}
    xo = x.clone().requires_grad_() if soft else x
    (O.cnn_forward(sdg, xo, sigma) * Gm).sum().backward()
    xg = x.to(dev).requires_grad_() if soft else x.to(dev)
    (m.backbone(xg, sigma.to(dev)) * Gm.to(dev)).sum().backward()
    worst = ('', 0.0)
    for name, p in m.named_parameters():
        ref = sdg[name].grad
        if ref is None:
            continue
        assert p.grad is not None and p.grad.shape == ref.shape, name
        e = (p.grad.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
        worst = max(worst, (name, e), key=lambda t: t[1])
        assert e < GTOL, (name, e)
    print('worst narrow-model gradient error:', worst)
    if soft:
        assert (xg.grad.cpu() - xo.grad).abs().max().item() / xo.grad.abs().max().item() < GTOL
    assert bsd  # (state dict keys carry the reference's 'backbone.' prefix)


def test_narrow_model_refuses_bf16_path(dev):
    m, _ = _model(dev, 32, steps=4)
    m.hip_dtype = 'bf16'
    with pytest.raises(NotImplementedError):
        m._sample(eval_sp_size=2, seed=1)


def test_narrow_bf16x3_tokens_equal_exact_path(dev):
    """hip_dtype = 'bf16x3' (fp32-grade convolutions on the bf16 matrix cores) takes the same zero-padded weights."""
    m, sd = _model(dev, 32, steps=24)
    want = m._sample(eval_sp_size=6, seed=11)
    m.hip_dtype = 'bf16x3'
    assert torch.equal(m._sample(eval_sp_size=6, seed=11), want)


def test_narrow_finetune_gradients_flat_and_per_parameter(dev):
    """The flat parameter-gradient vector of _sample_finetune_gradient on a narrow model (its gradients are slices of the
    padded ones, gathered per call) against the per-parameter form."""
    grads = {}
    for flat in (False, True):
        m, _ = _model(dev, 32, steps=10, truncate_steps=4)
        m.train()
        m.hip_flat_grads = flat
        rw = torch.randn(4, 200, generator=torch.Generator().manual_seed(5)).to(dev)
        out = m._sample_finetune_gradient(eval_sp_size=3, seed=7)[0]
        (out.permute(0, 2, 1) * rw).sum().backward()
        grads[flat] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    assert grads[False].keys() == grads[True].keys()
    for k, a in grads[False].items():
        b = grads[True][k]
        assert a.shape == b.shape == dict(m.named_parameters())[k].shape
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-3), k
