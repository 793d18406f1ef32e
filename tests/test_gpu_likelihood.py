"""Diffusion.q_xt / get_likelihood on the HIP path against fixture F7 (the reference's own methods, noise injected)."""
import numpy as np
import pytest
import torch

import drakes_amd
from conftest import load_golden
from oracle import philox

pytestmark = pytest.mark.gpu


def test_q_xt_exact(gpu_model, dev):
    g = load_golden('f7_likelihood.npz')
    x0 = torch.from_numpy(g['x0']).to(dev)
    seed, step, mc = int(g['useed']), int(g['qxt_step']), float(g['qxt_mc'])
    xt = gpu_model.q_xt(x0, torch.full((x0.shape[0], 1), mc, device=dev), seed=seed, step=step)      # in-kernel Philox
    assert torch.equal(xt.cpu(), torch.from_numpy(g['qxt']))
    u = torch.from_numpy(philox.step_uniforms(seed, step, *x0.shape)[:, :, 0].copy()).to(dev)
    assert torch.equal(gpu_model.q_xt(x0, mc, uniforms=u).cpu(), torch.from_numpy(g['qxt']))         # injected


@pytest.mark.parametrize('tag,tc', [('', False), ('_tc', True)])
def test_get_likelihood_golden(dev, det_sd, tag, tc):
    g = load_golden('f7_likelihood.npz')
    steps = int(g['steps' + tag])
    m = drakes_amd.Diffusion(drakes_amd.default_config(steps=steps, time_conditioning=tc)).to(dev)
    m.load_state_dict(det_sd)
    x0 = torch.from_numpy(g['x0']).to(dev)
    ll = m.get_likelihood(x0, n_samples=int(g['n_samples']), seed=int(g['useed'])).cpu().numpy()
    print(ll, g['loglik' + tag])
    np.testing.assert_allclose(ll, g['loglik' + tag], rtol=2e-5)
    # hip_dtype = 'bf16': the same loop with the denoiser on the bf16 layer kernels.  Floating-point bar: logits within
    # 6e-2 / 1e-2 rms of fp32 (tests/test_gpu_bf16.py), which leaves the summed log-likelihood within 1e-3 relative.
    m.hip_dtype = 'bf16'
    ll16 = m.get_likelihood(x0, n_samples=int(g['n_samples']), seed=int(g['useed'])).cpu().numpy()
    print('bf16:', ll16)
    np.testing.assert_allclose(ll16, g['loglik' + tag], rtol=1e-3)
