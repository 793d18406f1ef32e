"""The oracle against the golden vectors captured from the real reference
(oracle/make_golden.py).  CPU only.  On the host that generated the fixtures the match is
bit-exact; on another CPU torch may pick different SIMD kernels, so floats get an fp32 tolerance
while integer outputs must still be exact."""
import numpy as np
import torch

from conftest import load_golden
from oracle import drakes_oracle as O
from oracle import philox

RTOL, ATOL = 2e-5, 2e-5


def uni(seed):
    return lambda i, shp: torch.from_numpy(philox.step_uniforms(seed, i, shp[0], shp[1]))


def test_philox_known_answers():
    for ctr, key, expect in philox.KAT:
        r = philox.philox4x32_10(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert tuple(int(v) for v in r) == expect


def test_uniform_range_and_grid():
    u = philox.step_uniforms(1, 2, 16, 200)
    assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0
    assert np.all(u * 2 ** 24 == np.floor(u * 2 ** 24))
    # sharding invariance: rows 8.. of a batch of 16 == rows 0.. of a shard with offset 8
    assert np.array_equal(u[8:], philox.step_uniforms(1, 2, 8, 200, sample_offset=8))


def test_f1_step_scalars_bits():
    g = load_golden('f1_step_scalars.npz')
    tab = O.step_scalars(128).numpy().view(np.uint32)
    assert np.array_equal(tab, g['table_bits'])
    # SURVEY smoke values
    t = O.step_scalars(128)
    assert abs(t[0, 2].item() - 0.99900001) < 1e-7 and abs(t[127, 3].item() - 1.0013580e-05) < 1e-10


def test_f2_backbone(det_sd):
    g = load_golden('f2_backbone.npz')
    x = torch.from_numpy(g['x_int'])
    soft = torch.from_numpy(g['soft'])
    sigma = torch.from_numpy(g['sigma'])
    z = torch.zeros(4)
    with torch.no_grad():
        for got, key in ((O.cnn_forward(det_sd, x, z), 'logits_int'), (O.cnn_forward(det_sd, soft, z), 'logits_soft'),
                         (O.cnn_forward(det_sd, x, sigma), 'logits_int_tc'),
                         (O.cnn_forward(det_sd, soft, sigma), 'logits_soft_tc'),
                         (O.time_embedding(det_sd, sigma), 'time_emb')):
            np.testing.assert_allclose(got.numpy(), g[key], rtol=RTOL, atol=ATOL, err_msg=key)
        for got, key in ((O.model_forward(det_sd, x, sigma[:, None]), 'logp_int'),
                         (O.model_forward(det_sd, soft, sigma[:, None]), 'logp_soft'),
                         (O.model_forward(det_sd, x, sigma[:, None], True), 'logp_int_tc')):
            np.testing.assert_allclose(got.numpy(), g[key], rtol=RTOL, atol=ATOL, err_msg=key)


def test_f3_hard_step_exact():
    g = load_golden('f3_hard_step.npz')
    tab = O.step_scalars(128)
    for k in range(int(g['n'])):
        step = int(g[f'step_{k}'])
        logits = torch.from_numpy(g[f'logits_{k}'])
        x = torch.from_numpy(g[f'x_{k}'])
        u = torch.from_numpy(philox.step_uniforms(int(g['useed']), step, *x.shape))
        out = O.hard_step_from_logits(logits, x, u, tab[step, 2], tab[step, 3])
        assert torch.equal(out, torch.from_numpy(g[f'xnext_{k}'])), f'step {step}'
        # carry-over: unmasked positions never change
        assert torch.equal(out[x != 4], x[x != 4])


def test_f4_sample_short(det_sd):
    g = load_golden('f4_sample_short.npz')
    toks = O.sample(det_sd, 5, 64, 16, uni(int(g['useed'])))
    assert torch.equal(toks, torch.from_numpy(g['tokens']))
    assert toks.min() >= 0 and toks.max() <= 3


def test_f4_sample_full(det_sd):
    g = load_golden('f4_sample.npz')
    trace = []
    toks = O.sample(det_sd, 8, 200, 128, uni(int(g['useed'])), trace=trace)
    mism = (toks != torch.from_numpy(g['tokens'])).float().mean().item()
    assert mism == 0.0, f'token mismatch rate {mism}'
    assert torch.equal(trace[31], torch.from_numpy(g['state_after_32']))


def test_f5_finetune_gradient(det_sd):
    g = load_golden('f5_finetune_s16k6.npz')
    sd = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in det_sd.items()}
    steps, trunc = int(g['steps']), int(g['trunc'])
    out, lx, ct, mct, cf = O.sample_finetune_gradient(sd, 2, 200, steps, trunc, uni(int(g['useed'])),
                                                       temp=float(g['temp']))
    assert out.shape == (2, 200, 4) and out.requires_grad
    np.testing.assert_allclose(out.detach().numpy(), g['sample'], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(torch.stack([t.detach() for t in lx]).numpy(), g['last_x'], rtol=1e-4, atol=1e-5)
    assert cf[0].dtype == torch.int64 and cf[-1].dtype == torch.float32   # int64 then fp32
    assert np.array_equal(torch.stack(cf[:steps - trunc]).numpy(), g['copy_flag_hard'])
    np.testing.assert_allclose(torch.stack([t.detach() for t in cf[steps - trunc:]]).numpy(), g['copy_flag_soft'],
                               rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(ct).numpy(), g['condt'], rtol=1e-6)
    np.testing.assert_allclose(torch.stack(mct).numpy(), g['mc_t'], rtol=1e-6)
    rw = torch.from_numpy(g['reward_w'])
    (-(out.permute(0, 2, 1) * rw).sum((1, 2)).mean()).backward()
    for key in g.files:
        if key.startswith('grad_backbone'):
            name = key[len('grad_'):]
            got = sd[name].grad.numpy()
            ref = g[key]
            got = got[:ref.shape[0]]
            scale = max(np.abs(ref).max(), 1e-6)
            assert np.abs(got - ref).max() <= 1e-4 * scale, name
    gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.grad is not None)).item()
    assert abs(gn - float(g['grad_norm'])) <= 1e-4 * float(g['grad_norm'])


def test_f6_finetune_loop_oracle():
    """oracle/finetune_oracle.py against what the reference's own fine_tune produced (fixture F6 b: one optimizer step,
    truncated KL, soft copy flag)."""
    from types import SimpleNamespace as NS
    from oracle import finetune_oracle as FO
    from drakes_amd.det_init import det_state_dict
    g = load_golden('f6_finetune_loop_b.npz')
    lr = float(g['lr'])
    args = NS(truncate_steps=int(g['trunc']), gumbel_temp=float(g['temp']), total_num_steps=int(g['steps']), learning_rate=lr,
              num_epochs=int(g['epochs']), num_accum_steps=int(g['accum']), batch_size=int(g['batch']),
              copy_flag_temp=float(g['cft']), truncate_kl=bool(int(g['truncate_kl'])), alpha=float(g['alpha']),
              alpha_schedule_warmup=int(g['warmup']), gradnorm_clip=1.0)
    sd = det_state_dict(int(g['wseed']), 128, int(g['stacks']))
    gen = np.random.Generator(np.random.PCG64(int(g['old_perturb_seed'])))
    sd_old = {k: (v + 0.02 * torch.from_numpy(gen.standard_normal(tuple(v.shape)).astype(np.float32))
                  if k != 'backbone.time_embedder.0.W' else v.clone()) for k, v in sd.items()}
    sd_new = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in sd.items()}
    pnames = [k for k in sd if k != 'backbone.time_embedder.0.W']
    base = int(g['base_seed'])
    first = {}
    losses, log = FO.fine_tune(sd_new, sd_old, pnames, FO.SyntheticReward(3), FO.SyntheticReward(4), args, 200,
                               lambda i: base + i, record_first_grads=first)
    np.testing.assert_allclose(np.array([float(x) for x in losses], dtype=np.float32), g['losses'], rtol=1e-5)
    for key in g.files:
        if key.startswith('grad1_'):
            name = key[len('grad1_'):]
            ref = torch.from_numpy(g[key])
            assert float((first[name][:ref.shape[0]] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-9, name
        if key.startswith('delta_'):
            name = key[len('delta_'):]
            ref = torch.from_numpy(g[key])
            got = (sd_new[name].detach() - sd[name])[:ref.shape[0]]
            assert float((got - ref).abs().mean()) <= 1e-3 * lr, name


def test_f7_likelihood_oracle(det_sd):
    g = load_golden('f7_likelihood.npz')
    x0 = torch.from_numpy(g['x0'])
    B, L = x0.shape
    seed = int(g['useed'])
    u = lambda s, i: torch.from_numpy(philox.step_uniforms(seed + s, i, B, L)[:, :, 0].copy())
    ll = O.get_likelihood(det_sd, x0, int(g['steps_tc']), u, n_samples=int(g['n_samples']), time_conditioning=True)
    np.testing.assert_allclose(ll.numpy(), g['loglik_tc'], rtol=1e-6)
    assert torch.equal(O.q_xt(x0, float(g['qxt_mc']), torch.from_numpy(philox.step_uniforms(seed, int(g['qxt_step']), B, L)[:, :, 0].copy())),
                       torch.from_numpy(g['qxt']))


def test_f8_protein_step_oracle():
    g = load_golden('f8_protein_sample.npz')
    table, traj, clean = torch.from_numpy(g['table']), torch.from_numpy(g['traj']), torch.from_numpy(g['clean'])
    T = int(g['T'])
    ts = torch.linspace(float(g['min_t']), 1.0, T)
    x, t_1 = traj[0], ts[0]
    B, L, V = table.shape[1:]
    for k, t_2 in enumerate(ts[1:]):
        u = torch.from_numpy(philox.step_uniforms_wide(int(g['useed']), k, B, L, V))
        x, pc = O.hard_step_wide(table[k], x, u, t_2 - t_1, 1.0 - t_2, temp=float(g['temp']))
        assert torch.equal(x, traj[k + 1]) and torch.equal(pc, clean[k])
        t_1 = t_2


def test_f9_protein_sample_gradient_oracle():
    g = load_golden('f9_protein_sample_gradient.npz')
    tab = torch.from_numpy(g['table']).requires_grad_()
    mix, W = torch.from_numpy(g['mix']), torch.from_numpy(g['W'])
    B, L, V = tab.shape[1:]
    out, lx, mct, cf = O.sample_gradient_wide(lambda k, xs: tab[k] + xs @ mix, B, L, int(g['T']), float(g['min_t']),
                                              float(g['temp']), int(g['K']), float(g['gumbel_temp']),
                                              lambda k: torch.from_numpy(philox.step_uniforms_wide(int(g['useed']), k, B, L, V)))
    (out * W).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), g['sample'], rtol=1e-6, atol=1e-7)
    assert float((tab.grad - torch.from_numpy(g['grad_table'])).abs().max()) <= 1e-5 * float(np.abs(g['grad_table']).max())
