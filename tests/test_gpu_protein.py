"""The protein sampler's reverse loop (V = 22) on the HIP step against fixture F8: the reference's own
Interpolant.sample (fm_utils.py:86-139) with a table in place of ProteinMPNN and Philox noise injected."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from conftest import load_golden
from drakes_amd import ops, protein
from oracle import drakes_oracle as O
from oracle import philox

pytestmark = pytest.mark.gpu


def test_uniforms_wide_match_oracle_bits(dev):
    """In-kernel wide Philox stream == oracle: a step whose logits make the draw decisive reproduces the oracle tokens."""
    g = np.random.Generator(np.random.PCG64(2))
    B, L, V = 5, 93, 22
    logits = torch.from_numpy(g.standard_normal((B, L, V)).astype(np.float32) * 0.05)     # near-uniform: noise decides
    x = torch.full((B, L), 21, dtype=torch.int64)
    x[1, ::3] = torch.from_numpy(g.integers(0, 21, (L + 2) // 3))
    for step, off in ((0, 0), (7, 1000), (3, 2 ** 33 + 5)):
        u = torch.from_numpy(philox.step_uniforms_wide(99, step, B, L, V, off))
        want, clean = O.hard_step_wide(logits, x, u, 0.11, 0.52, temp=0.9)
        got, gclean = ops.reverse_step_hard_wide(logits.to(dev), x.to(dev), 0.11, 0.52, 0.9, 21, 99, step, off)
        assert torch.equal(got.cpu(), want) and torch.equal(gclean.cpu(), clean)
        got2, _ = ops.reverse_step_hard_wide(logits.to(dev), x.to(dev), 0.11, 0.52, 0.9, 21, uniforms=u.to(dev))
        assert torch.equal(got2.cpu(), want)


def test_interpolant_sample_golden(dev):
    g = load_golden('f8_protein_sample.npz')
    table = torch.from_numpy(g['table']).to(dev)
    T = int(g['T'])
    it = protein.Interpolant(NS(num_timesteps=T, min_t=float(g['min_t']), temp=float(g['temp']), interpolant_type='masking'))
    it.set_device(dev)
    calls = {'k': 0}

    def model(X, S, mask, chain_M, residue_idx, chain_encoding_all, cls=None):
        assert S.dtype == torch.int64 and S.is_cuda
        out = table[calls['k']]
        calls['k'] += 1
        return out
    mask = torch.ones(table.shape[1], table.shape[2], device=dev)
    pred, traj, clean = it.sample(model, None, mask, mask, None, None, seed=int(g['useed']))
    assert len(traj) == T and len(clean) == T - 1
    assert np.array_equal(torch.stack(traj).numpy(), g['traj'])            # every intermediate state, bit-exact
    assert np.array_equal(torch.stack(clean).numpy(), g['clean'])
    assert torch.equal(pred.cpu(), clean[-1])


def test_wide_step_rejects_other_vocabularies(dev):
    from drakes_amd import DrakesHipError
    with pytest.raises(DrakesHipError):
        ops.reverse_step_hard_wide(torch.zeros(1, 4, 20, device=dev), torch.zeros(1, 4, dtype=torch.int64, device=dev), 0.1, 0.5)


def test_interpolant_sample_gradient_golden(dev):
    """Interpolant.sample_gradient (fm_utils.py:142-209) against fixture F9: the reference's own method with a
    differentiable stand-in for ProteinMPNN (table[k] + x @ mix), gradients taken w.r.t. the table."""
    g = load_golden('f9_protein_sample_gradient.npz')
    T, K = int(g['T']), int(g['K'])
    table = torch.from_numpy(g['table']).to(dev).requires_grad_()
    mix, W = torch.from_numpy(g['mix']).to(dev), torch.from_numpy(g['W']).to(dev)
    it = protein.Interpolant(NS(num_timesteps=T, min_t=float(g['min_t']), temp=float(g['temp']), interpolant_type='masking'))
    it.set_device(dev)
    calls = {'k': 0}

    def model(X, S, mask, chain_M, residue_idx, chain_encoding_all, cls=None):
        assert S.shape[-1] == 22 and S.dtype == torch.float32
        out = table[calls['k']] + S @ mix
        calls['k'] += 1
        return out
    mask = torch.ones(table.shape[1], table.shape[2], device=dev)
    out, lx, mct, cf = it.sample_gradient(model, None, mask, mask, None, None, K, float(g['gumbel_temp']), seed=int(g['useed']))
    assert len(lx) == len(mct) == len(cf) == T and out.requires_grad
    np.testing.assert_allclose(out.detach().cpu().numpy(), g['sample'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lx[-1].detach().cpu().numpy(), g['last_x_final'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(lx[T - K + 1].detach().cpu().numpy(), g['last_x_first_soft'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(cf[-1].detach().cpu().numpy(), g['copy_flag_final'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(np.array([float(m) for m in mct], dtype=np.float32), g['mct'], rtol=1e-6)
    (out * W).sum().backward()
    ref = torch.from_numpy(g['grad_table'])
    got = table.grad.cpu()
    assert float(got[: T - K].abs().max()) == 0.0                       # hard steps carry no gradient
    e = float((got - ref).abs().max() / ref.abs().max())
    print('gradient error vs reference:', e)
    assert e < 1e-4
