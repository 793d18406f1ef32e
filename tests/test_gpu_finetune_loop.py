"""The fine-tune loop on the HIP path against fixtures F6, which hold what the REFERENCE's own
`fine_tune` function produced (oracle/make_golden_finetune.py) on the same weights, reward model and injected noise."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

import drakes_amd
from conftest import load_golden
from drakes_amd import autograd
from drakes_amd.det_init import det_state_dict
from oracle import drakes_oracle as O
from oracle.finetune_oracle import SyntheticReward

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('soft', [False, True])
def test_kl_step_forward_backward(dev, soft):
    """One fused kernel each way vs torch autograd on the reference formula (finetune_reward_bp.py:70-74)."""
    gen = np.random.Generator(np.random.PCG64(8))
    B, L = 3, 200
    lp = torch.log_softmax(torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32)), -1)
    lpo = torch.log_softmax(torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32)), -1)
    if soft:
        x = torch.from_numpy(gen.dirichlet(np.ones(5) * 0.5, (B, L)).astype(np.float32))
        flag = torch.from_numpy(gen.random((B, L, 1)).astype(np.float32))
    else:
        x = torch.nn.functional.one_hot(torch.from_numpy(gen.integers(0, 5, (B, L))), 5).float()
        flag = torch.from_numpy(gen.integers(0, 2, (B, L, 1)))
    G = torch.from_numpy(gen.standard_normal(B).astype(np.float32))
    mc = 0.3
    a, xo, fo = lp.clone().requires_grad_(), x.clone().requires_grad_(soft), flag.clone().requires_grad_(soft) if soft else flag
    lo = lpo.clone().requires_grad_(soft)
    want = O.kl_step(a, lo, xo, fo, mc)
    (want * G).sum().backward()
    b, xg = lp.to(dev).requires_grad_(), x.to(dev).requires_grad_(soft)
    fg = flag.to(dev).requires_grad_(soft) if soft else flag.to(dev)
    lg = lpo.to(dev).requires_grad_(soft)
    got = autograd.kl_step_with_grad(b, lg, xg, fg, mc)
    (got * G.to(dev)).sum().backward()
    rel = lambda u, v: float((u.cpu() - v).abs().max() / v.abs().max().clamp_min(1e-12))
    assert rel(got.detach(), want.detach()) < 1e-5
    assert rel(b.grad, a.grad) < 1e-5 and float(b.grad[..., 4].abs().max()) == 0.0
    if soft:
        assert rel(xg.grad, xo.grad) < 1e-5 and rel(fg.grad, fo.grad) < 1e-5 and rel(lg.grad, lo.grad) < 1e-5


@pytest.mark.timeout(600)
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_fine_tune_matches_reference(dev, tag):
    g = load_golden(f'f6_finetune_loop_{tag}.npz')
    steps, trunc, stacks, lr = int(g['steps']), int(g['trunc']), int(g['stacks']), float(g['lr'])
    cft = None if float(g['cft']) < 0 else float(g['cft'])
    args = NS(truncate_steps=trunc, gumbel_temp=float(g['temp']), total_num_steps=steps, learning_rate=lr,
              num_epochs=int(g['epochs']), num_accum_steps=int(g['accum']), batch_size=int(g['batch']), copy_flag_temp=cft,
              truncate_kl=bool(int(g['truncate_kl'])), alpha=float(g['alpha']), alpha_schedule_warmup=int(g['warmup']),
              gradnorm_clip=1.0, name='debug', save_every_n_epochs=10 ** 9)
    sd = det_state_dict(int(g['wseed']), 128, stacks)
    gen = np.random.Generator(np.random.PCG64(int(g['old_perturb_seed'])))
    sd_old = {k: (v + 0.02 * torch.from_numpy(gen.standard_normal(tuple(v.shape)).astype(np.float32))
                  if k != 'backbone.time_embedder.0.W' else v.clone()) for k, v in sd.items()}

    def model(state):
        m = drakes_amd.Diffusion(drakes_amd.default_config(steps=steps, num_cnn_stacks=stacks)).to(dev)
        m.load_state_dict(state)
        return m
    new_model, old_model = model(sd), model(sd_old)
    reward, reward_eval = SyntheticReward(3).to(dev), SyntheticReward(4).to(dev)
    base = int(g['base_seed'])
    first = {}

    def grab(model, epoch):
        if not first:
            first.update({k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None})
    losses = drakes_amd.fine_tune(new_model, reward, reward_eval, old_model, args, seed_of_call=lambda i: base + i,
                                  verbose=False, on_update=grab)
    losses = np.array([float(x) for x in losses], dtype=np.float32)
    print('losses', losses, 'reference', g['losses'])
    # before the first optimizer step the weights are the reference's: losses agree to kernel precision
    n0 = int(g['accum'])
    np.testing.assert_allclose(losses[:n0], g['losses'][:n0], rtol=2e-4, atol=2e-5)
    # afterwards they depend on the Adam updates (checked below) AND on discrete samples: the hard steps draw tokens by
    # argmax, so a weight difference at rounding level can flip a token and move the reward by O(1/L); loose bound
    np.testing.assert_allclose(losses, g['losses'], rtol=0, atol=1e-2)
    # the accumulated gradient of the first optimizer step (reward + alpha * KL over all recorded steps, unclipped)
    for key in g.files:
        if key.startswith('grad1_'):
            name = key[len('grad1_'):]
            ref = torch.from_numpy(g[key])
            got = first[name][:ref.shape[0]]
            e = float((got - ref).abs().max() / ref.abs().max())
            print(name, 'first-step gradient error', e)
            assert e < 5e-3, (name, e)
    # Adam's first step moves every weight by lr * g / (|g| + 1e-8), about lr * sign(gradient): the weights must move the
    # reference's way wherever the gradient is not within rounding of zero.  Later steps (fixture c) start from tokens
    # sampled by argmax from already-updated weights, so only a loose bound holds there.
    one_step = int(g['epochs']) == 1
    final = new_model.state_dict()
    for key in g.files:
        if key.startswith('delta_'):
            name = key[len('delta_'):]
            ref = torch.from_numpy(g[key])
            got = (final[name].cpu() - sd[name])[:ref.shape[0]]
            d = (got - ref).abs()
            frac_far = float((d > 0.1 * lr).float().mean())
            print(name, 'mean |d|/lr', float(d.mean()) / lr, 'fraction beyond 0.1 lr', frac_far)
            assert float(d.mean()) <= (0.01 if one_step else 0.05) * lr, (name, float(d.mean()) / lr)
            assert frac_far <= (0.01 if one_step else 0.05), (name, frac_far)
    if one_step:
        assert abs(drakes_amd.fine_tune.last_log['tot_grad_norm'] - float(g['grad_norms'][0])) <= 2e-3 * float(g['grad_norms'][0])
        assert abs(drakes_amd.fine_tune.last_log['mean_kl'] - float(g['kl_losses'][0])) <= 2e-3 * abs(float(g['kl_losses'][0])) + 1e-5


def test_hard_step_kl_is_exactly_zero(dev, det_sd):
    """What `skip_hard_step_kl` relies on: with and without it the losses and the accumulated gradients are bit-identical
    (the weight-gradient kernels use atomics, so the comparison runs on a loss whose hard-step KL would be the only
    difference: truncate_kl=False, reward weight zero is not available -> compare the KL values themselves)."""
    from types import SimpleNamespace as NS
    from drakes_amd import finetune_loop
    m = drakes_amd.Diffusion(drakes_amd.default_config(steps=10, truncate_steps=3)).to(dev)
    m.load_state_dict(det_sd)
    old = drakes_amd.Diffusion(drakes_amd.default_config(steps=10)).to(dev)
    old.load_state_dict({k: v + 0.01 for k, v in det_sd.items()})
    m.train()
    sample, lx, ct, mct, cf = m._sample_finetune_gradient(eval_sp_size=3, seed=12)
    for t in range(7):                                   # the hard steps
        assert cf[t].dtype == torch.int64
        m.zero_grad()
        kl = finetune_loop.kl_of_step(m, old, lx[t], ct[t], mct[t], cf[t])
        assert float(kl.detach().abs().max()) == 0.0
        kl.sum().backward()
        assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in m.parameters())
    kl_soft = finetune_loop.kl_of_step(m, old, lx[8], ct[8], mct[8], cf[8])
    assert float(kl_soft.detach().abs().max()) > 0.0


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_kl_side_streams_change_nothing(dev, det_sd, dtype):
    """fine_tune(kl_streams=2) runs the frozen model's KL forwards (and, through autograd, their input-gradient
    launches) on two side streams: losses and the first optimizer step's gradients equal those of the one-stream loop
    (the same kernels on the same data; only the fp32 paths with atomics may differ in the last bits)."""
    args = NS(truncate_steps=4, gumbel_temp=1.0, total_num_steps=12, learning_rate=1e-4, num_epochs=1, num_accum_steps=2,
              batch_size=3, copy_flag_temp=None, truncate_kl=False, alpha=0.01, alpha_schedule_warmup=0, gradnorm_clip=1.0,
              name='debug', save_every_n_epochs=10 ** 9)
    res = {}
    for ks in (0, 2):
        new_model = drakes_amd.Diffusion(drakes_amd.default_config(steps=12)).to(dev)
        old_model = drakes_amd.Diffusion(drakes_amd.default_config(steps=12)).to(dev)
        new_model.load_state_dict(det_sd)
        old_model.load_state_dict({k: v + 0.01 for k, v in det_sd.items()})
        new_model.hip_dtype = old_model.hip_dtype = dtype
        reward, reward_eval = SyntheticReward(3).to(dev), SyntheticReward(4).to(dev)
        first = {}

        def grab(model, epoch):
            first.update({k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None})
        losses = drakes_amd.fine_tune(new_model, reward, reward_eval, old_model, args, seed_of_call=lambda i: 50 + i,
                                      verbose=False, on_update=grab, kl_streams=ks)
        res[ks] = ([float(x) for x in losses], first)
    np.testing.assert_allclose(res[0][0], res[2][0], rtol=1e-6, atol=1e-7)
    for k, g0 in res[0][1].items():
        g2 = res[2][1][k]
        scale = float(g0.abs().max()) + 1e-12
        assert float((g0 - g2).abs().max()) <= 1e-5 * scale, (k, float((g0 - g2).abs().max()) / scale)
