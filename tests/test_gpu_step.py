"""Reverse-step kernels against the oracle and the reference's golden vectors.  Bit-exact:
token ids and uniforms are integer-valued results."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from drakes_amd import ops
from oracle import drakes_oracle as O
from oracle import philox

pytestmark = pytest.mark.gpu


def test_uniforms_match_philox_bits(dev):
    for (B, L, seed, step, off) in ((3, 200, 2024, 0, 0), (5, 64, 2 ** 40 + 17, 127, 0), (2, 1024, 9, 511, 2 ** 33 + 5),
                                    (1, 1, 0, 0, 0)):
        u = ops.step_uniforms(B, L, seed, step, off, device=dev).cpu().numpy()
        ref = philox.step_uniforms(seed, step, B, L, off)
        assert np.array_equal(u.view(np.uint32), ref.view(np.uint32))


def test_hard_step_golden_exact(dev):
    """(logits, x, U, mc_t, mc_s) -> x_next captured from the reference's _ddpm_update."""
    g = load_golden('f3_hard_step.npz')
    tab = O.step_scalars(128)
    seed = int(g['useed'])
    for k in range(int(g['n'])):
        step = int(g[f'step_{k}'])
        logits = torch.from_numpy(g[f'logits_{k}']).to(dev)
        x = torch.from_numpy(g[f'x_{k}']).to(dev)
        want = torch.from_numpy(g[f'xnext_{k}'])
        mc_t, mc_s = tab[step, 2].item(), tab[step, 3].item()
        got_rng = ops.reverse_step_hard(logits, x, mc_t, mc_s, seed, step, 0).cpu()
        u = torch.from_numpy(philox.step_uniforms(seed, step, *x.shape)).to(dev)
        got_inj = ops.reverse_step_hard(logits, x, mc_t, mc_s, uniforms=u).cpu()
        assert torch.equal(got_rng, want), f'in-kernel RNG, step {step}'
        assert torch.equal(got_inj, want), f'injected uniforms, step {step}'


@pytest.mark.parametrize('B,L', [(1, 1), (3, 7), (64, 200), (16, 1024)])
def test_hard_step_vs_oracle_random(dev, B, L):
    gen = np.random.Generator(np.random.PCG64(B * 1000 + L))
    tab = O.step_scalars(128)
    total = mism = 0
    for step in (0, 37, 127):
        logits = torch.from_numpy((gen.standard_normal((B, L, 5)) * 3).astype(np.float32))
        x = torch.from_numpy(gen.integers(0, 5, (B, L)))
        x[:, : L // 2] = 4
        u = torch.from_numpy(philox.step_uniforms(5, step, B, L, sample_offset=11))
        want = O.hard_step_from_logits(logits, x, u, tab[step, 2], tab[step, 3])
        got = ops.reverse_step_hard(logits.to(dev), x.to(dev), tab[step, 2].item(), tab[step, 3].item(), 5, step, 11)
        mism += int((got.cpu() != want).sum())
        total += B * L
    assert mism == 0, f'{mism}/{total} tokens differ'


def test_hard_step_in_place_and_empty(dev):
    tab = O.step_scalars(128)
    logits = torch.randn(4, 50, 5, device=dev)
    x = torch.full((4, 50), 4, dtype=torch.int64, device=dev)
    ref = ops.reverse_step_hard(logits, x, tab[5, 2].item(), tab[5, 3].item(), 1, 5, 0)
    ops.reverse_step_hard(logits, x, tab[5, 2].item(), tab[5, 3].item(), 1, 5, 0, out=x)
    assert torch.equal(ref, x)
    e = ops.reverse_step_hard(torch.empty(0, 50, 5, device=dev), torch.empty(0, 50, dtype=torch.int64, device=dev), .5, .4)
    assert e.shape == (0, 50)


def test_subs_fwd_int_and_soft(dev):
    gen = np.random.Generator(np.random.PCG64(3))
    logits = torch.from_numpy((gen.standard_normal((6, 200, 5)) * 4).astype(np.float32))
    x = torch.from_numpy(gen.integers(0, 5, (6, 200)))
    soft = torch.from_numpy(gen.dirichlet(np.ones(5) * 0.3, (6, 200)).astype(np.float32))
    soft[0, :5] = torch.tensor([0.5, 0.5, 0., 0., 0.])      # argmax tie -> first index, like torch
    for xt in (x, soft):
        want = O.subs_parameterization(logits.clone(), xt)
        got = ops.subs_fwd(logits.to(dev), xt.to(dev)).cpu()
        # tolerance: fp32 exp/log differ by <= 2 ulp between libm flavours; |logp| <= ~30 here
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=4e-6)
        fixed = (want == 0) | (want == -1000000.0)
        assert torch.equal(got[fixed], want[fixed])


def test_denoise_argmax(dev):
    gen = np.random.Generator(np.random.PCG64(4))
    logits = torch.from_numpy((gen.standard_normal((8, 200, 5)) * 4).astype(np.float32))
    x = torch.from_numpy(gen.integers(0, 5, (8, 200)))
    want = O.subs_parameterization(logits.clone(), x)[:, :, :-1].argmax(-1)
    got = ops.denoise_argmax(logits.to(dev), x.to(dev)).cpu()
    assert torch.equal(got, want)
    assert int(got.max()) <= 3


def test_hard_step_per_row_timesteps(dev, gpu_model):
    """_ddpm_update accepts any t [B,1] (diffusion_gosai_update.py:462-486): rows at different times get their own
    (move_chance_t, move_chance_s).  Kernel vs oracle (bit-exact), then the method itself."""
    gen = np.random.Generator(np.random.PCG64(91))
    B, L = 7, 200
    logits = torch.from_numpy((gen.standard_normal((B, L, 5)) * 3).astype(np.float32))
    x = torch.from_numpy(gen.integers(0, 5, (B, L)))
    x[:, ::2] = 4
    t = torch.tensor([1.0, 0.9, 0.61, 0.5, 0.33, 0.1, 0.02]).reshape(B, 1)
    dt = (1 - 1e-5) / 128
    mc_t = 1 - torch.exp(-O.total_noise(t).squeeze(-1))
    mc_s = 1 - torch.exp(-O.total_noise(t - dt).squeeze(-1))
    u = torch.from_numpy(philox.step_uniforms(5, 9, B, L))
    want = O.hard_step_from_logits(logits, x, u, mc_t[:, None, None], mc_s[:, None, None])
    rows = torch.stack([mc_t, mc_s], -1).contiguous().to(dev)
    for inj in (None, u.to(dev)):
        got = ops.reverse_step_hard(logits.to(dev), x.to(dev), 0.0, 0.0, 5, 9, 0, inj, mc_rows=rows).cpu()
        assert torch.equal(got, want)
    # all rows equal through mc_rows == the scalar form
    same = torch.tensor([[0.5, 0.4]] * B, device=dev)
    assert torch.equal(ops.reverse_step_hard(logits.to(dev), x.to(dev), 0.0, 0.0, 5, 9, 0, mc_rows=same),
                       ops.reverse_step_hard(logits.to(dev), x.to(dev), 0.5, 0.4, 5, 9, 0))
    # the drop-in method: mixed t, time conditioning off -> logits from the model, step vs oracle on those logits
    xd = x.to(dev)
    got = gpu_model._ddpm_update(xd, t.to(dev), dt, seed=5, step=9).cpu()
    with torch.no_grad():
        lg = gpu_model.backbone(xd, torch.zeros(B, device=dev)).cpu()
    assert torch.equal(got, O.hard_step_from_logits(lg, x, u, mc_t[:, None, None], mc_s[:, None, None]))
    with pytest.raises(ValueError):
        ops.reverse_step_hard(logits.to(dev), x.to(dev), 0.5, 0.4, mc_rows=rows[:3])


def test_q_xt_per_row_move_chance(dev, gpu_model):
    gen = np.random.Generator(np.random.PCG64(92))
    B, L = 5, 200
    x0 = torch.from_numpy(gen.integers(0, 4, (B, L)))
    mc = torch.tensor([0.05, 0.3, 0.5, 0.77, 0.999]).reshape(B, 1)
    u = torch.from_numpy(philox.step_uniforms(12, 3, B, L)[:, :, 0].copy())
    want = O.q_xt(x0, mc, u)
    got = gpu_model.q_xt(x0.to(dev), mc.to(dev), seed=12, step=3).cpu()
    assert torch.equal(got, want)
    got_inj = gpu_model.q_xt(x0.to(dev), mc.to(dev), uniforms=u.to(dev)).cpu()
    assert torch.equal(got_inj, want)
