"""Gumbel-softmax reverse step, denoiser backward and _sample_finetune_gradient against the
oracle (torch autograd on the pinned CPU restatement) and the reference's golden vectors.

Floating-point bars (fp32): forward values 2e-4 abs / 1e-4 rel; gradients 2e-3 of the tensor's
largest |gradient| (parameter gradients are sums over up to B*L*steps terms accumulated with fp32
atomics in a different order than torch's), global gradient norm 1e-3 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import drakes_amd
from conftest import load_golden
from drakes_amd import autograd, ops
from oracle import drakes_oracle as O
from oracle import philox

pytestmark = pytest.mark.gpu
GTOL = 2e-3


def relerr(got, ref):
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


@pytest.mark.parametrize('cft', [None, 0.5])
def test_soft_step_forward_backward(dev, cft):
    gen = np.random.Generator(np.random.PCG64(21))
    B, L, temp = 5, 200, 0.8
    logits = torch.from_numpy((gen.standard_normal((B, L, 5)) * 2).astype(np.float32))
    x = torch.from_numpy(gen.dirichlet(np.ones(5) * 0.4, (B, L)).astype(np.float32))
    x[0] = F.one_hot(torch.full((L,), 4), 5).float()               # fully masked rows (first soft step)
    x[1] = F.one_hot(torch.from_numpy(gen.integers(0, 4, L)), 5).float()   # fully unmasked one-hot rows
    u = torch.from_numpy(philox.step_uniforms(3, 9, B, L))
    tab = O.step_scalars(128)
    mc_t, mc_s = tab[100, 2], tab[100, 3]
    G = torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32))
    Gf = torch.from_numpy(gen.standard_normal((B, L, 1)).astype(np.float32))
    lo, xo = logits.clone().requires_grad_(), x.clone().requires_grad_()
    xn, fl = O.soft_step_from_logp(O.subs_parameterization(lo, xo), xo, u, mc_t, mc_s, temp, cft)
    ((xn * G).sum() + (fl * Gf).sum()).backward()
    lg, xg = logits.to(dev).requires_grad_(), x.to(dev).requires_grad_()
    for inject in (True, False):
        lg.grad = xg.grad = None
        xn2, fl2 = autograd.soft_step_with_grad(lg, xg, mc_t.item(), mc_s.item(), temp, cft, 3, 9, 0,
                                                u.to(dev) if inject else None)
        ((xn2 * G.to(dev)).sum() + (fl2 * Gf.to(dev)).sum()).backward()
        np.testing.assert_allclose(xn2.detach().cpu().numpy(), xn.detach().numpy(), rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(fl2.detach().cpu().numpy(), fl.detach().numpy(), rtol=1e-5, atol=1e-6)
        assert torch.isfinite(lg.grad).all() and torch.isfinite(xg.grad).all()
        assert relerr(lg.grad.cpu(), lo.grad) < 1e-4, relerr(lg.grad.cpu(), lo.grad)
        assert relerr(xg.grad.cpu(), xo.grad) < 1e-4
        assert float(lg.grad[1].abs().max()) == 0.0          # rows holding a token: no gradient into the logits


def test_soft_step_per_row_timesteps(dev):
    """_ddpm_update_finetune_gradient with rows at different times (reference :552-586 takes any t [B,1]): the kernels
    read one (mc_t, mc_s) pair per row; forward and backward against oracle autograd."""
    gen = np.random.Generator(np.random.PCG64(23))
    B, L, temp = 4, 200, 1.0
    logits = torch.from_numpy((gen.standard_normal((B, L, 5)) * 2).astype(np.float32))
    x = torch.from_numpy(gen.dirichlet(np.ones(5) * 0.4, (B, L)).astype(np.float32))
    u = torch.from_numpy(philox.step_uniforms(3, 9, B, L))
    t = torch.tensor([0.95, 0.6, 0.3, 0.05]).reshape(B, 1)
    dt = (1 - 1e-5) / 128
    mc_t = 1 - torch.exp(-O.total_noise(t).squeeze(-1))
    mc_s = 1 - torch.exp(-O.total_noise(t - dt).squeeze(-1))
    G = torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32))
    lo, xo = logits.clone().requires_grad_(), x.clone().requires_grad_()
    xn, fl = O.soft_step_from_logp(O.subs_parameterization(lo, xo), xo, u, mc_t[:, None, None], mc_s[:, None, None], temp, None)
    (xn * G).sum().backward()
    lg, xg = logits.to(dev).requires_grad_(), x.to(dev).requires_grad_()
    rows = torch.stack([mc_t, mc_s], -1).contiguous().to(dev)
    xn2, fl2 = autograd.soft_step_with_grad(lg, xg, 0.0, 0.0, temp, None, 3, 9, 0, None, rows)
    (xn2 * G.to(dev)).sum().backward()
    np.testing.assert_allclose(xn2.detach().cpu().numpy(), xn.detach().numpy(), rtol=1e-4, atol=2e-6)
    assert relerr(lg.grad.cpu(), lo.grad) < 1e-4 and relerr(xg.grad.cpu(), xo.grad) < 1e-4


def _oracle_grads(sd, fn):
    sdg = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in sd.items()}
    loss = fn(sdg)
    loss.backward()
    return sdg


def _check_param_grads(model, sdg, tol=GTOL):
    worst = ('', 0.0)
    for name, p in model.named_parameters():
        ref = sdg[name].grad
        if ref is None:
            assert p.grad is None or not p.requires_grad
            continue
        assert p.grad is not None, name
        e = relerr(p.grad.cpu(), ref)
        worst = max(worst, (name, e), key=lambda t: t[1])
        assert e < tol, (name, e)
    return worst


def _input_away_from_relu_kinks(sd, B, L, soft, sigma, first_seed):
    """ReLU's derivative jumps at 0: an argument within rounding distance of 0 can get a different sign on the GPU
    (accumulation order) and then a whole gradient term appears or disappears.  That is a property of the function,
    not an error, so the unit test differentiates at an input whose ReLU arguments all clear the kink by 5e-6
    (GPU and CPU pre-activations agree to ~1e-6)."""
    for seed in range(first_seed, first_seed + 400):
        gen = np.random.Generator(np.random.PCG64(seed))
        x = torch.from_numpy(gen.dirichlet(np.ones(5) * 0.5, (B, L)).astype(np.float32)) if soft else \
            torch.from_numpy(gen.integers(0, 5, (B, L)))
        pre = []
        with torch.no_grad():
            O.cnn_forward(sd, x, sigma, preacts=pre)
        if min(float(c.abs().min()) for c in pre) > 5e-6:
            return x, gen
    raise AssertionError('no input found away from the ReLU kinks')


@pytest.mark.parametrize('soft,per_row_sigma', [(False, False), (True, True), (True, False)])
def test_backbone_backward(dev, det_sd, soft, per_row_sigma):
    B, L = 2, 45
    m = drakes_amd.Diffusion(drakes_amd.default_config(length=L, time_conditioning=per_row_sigma)).to(dev)
    m.load_state_dict(det_sd)
    sigma = torch.tensor([0.3, 4.0]) if per_row_sigma else torch.zeros(B)
    x, gen = _input_away_from_relu_kinks(det_sd, B, L, soft, sigma, 5)
    Gm = torch.from_numpy(gen.standard_normal((B, L, 5)).astype(np.float32))
    xo = x.clone().requires_grad_() if soft else x
    sdg = _oracle_grads(det_sd, lambda sd: (O.cnn_forward(sd, xo, sigma) * Gm).sum())
    xg = x.to(dev).requires_grad_() if soft else x.to(dev)
    logits = m.backbone(xg, sigma.to(dev))
    assert logits.requires_grad
    (logits * Gm.to(dev)).sum().backward()
    worst = _check_param_grads(m, sdg)
    print('worst parameter gradient error:', worst)
    if soft:
        assert relerr(xg.grad.cpu(), xo.grad) < GTOL


def test_forward_logp_with_grad_kl_term(gpu_model, dev, det_sd):
    """Diffusion.forward under autograd, as the KL loop of fine_tune uses it (finetune_reward_bp.py:67-74)."""
    gen = np.random.Generator(np.random.PCG64(6))
    B, L = 2, 200
    last_x = F.one_hot(torch.from_numpy(gen.integers(0, 5, (B, L))), 5).float()
    condt = torch.full((B,), 0.69215769)
    old = {k: v + 0.01 * torch.from_numpy(gen.standard_normal(tuple(v.shape)).astype(np.float32)) for k, v in det_sd.items()}
    copy_flag = 1 - last_x[:, :, 4:5]
    with torch.no_grad():
        lp_old = O.model_forward(old, last_x, condt)
    sdg = _oracle_grads(det_sd, lambda sd: O.kl_step(O.model_forward(sd, last_x, condt), lp_old, last_x, copy_flag, 0.5).mean())
    gpu_model.zero_grad()
    lp = gpu_model.forward(last_x.to(dev), condt.to(dev))
    lpo, lx, cf = lp_old.to(dev), last_x.to(dev), copy_flag.to(dev)
    p, po = lp[:, :, :-1].exp(), lpo[:, :, :-1].exp()
    kl = (cf * (-p + po + p * (lp[:, :, :-1] - lpo[:, :, :-1])) / 0.5 * lx[:, :, :-1]).sum((1, 2)).mean()
    kl.backward()
    print('worst:', _check_param_grads(gpu_model, sdg))
    gpu_model.zero_grad()


@pytest.mark.parametrize('tag', ['s16k6', 's16k6_cft', 's128k50'])
def test_sample_finetune_gradient_golden(dev, det_sd, tag):
    g = load_golden(f'f5_finetune_{tag}.npz')
    steps, trunc, temp, cft = int(g['steps']), int(g['trunc']), float(g['temp']), float(g['cft'])
    cft = None if cft < 0 else cft
    m = drakes_amd.Diffusion(drakes_amd.default_config(steps=steps)).to(dev)
    m.load_state_dict(det_sd)
    m.train()
    m.config.finetuning.truncate_steps = trunc            # assigned by the caller, finetune_reward_bp.py:22-23
    m.config.finetuning.gumbel_softmax_temp = temp
    out, lx, ct, mct, cf = m._sample_finetune_gradient(eval_sp_size=2, copy_flag_temp=cft, seed=int(g['useed']))
    assert out.shape == (2, 200, 4) and out.dtype == torch.float32 and out.requires_grad
    assert len(lx) == len(ct) == len(mct) == len(cf) == steps
    assert lx[0].shape == (2, 200, 5) and ct[0].shape == (2,) and mct[0].shape == (2, 1, 1) and cf[0].shape == (2, 200, 1)
    assert cf[0].dtype == torch.int64 and cf[-1].dtype == torch.float32
    np.testing.assert_allclose(out.detach().cpu().numpy(), g['sample'], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(lx[-1].detach().cpu().numpy(), g['last_x_final'], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(cf[-1].detach().cpu().numpy(), g['copy_flag_final'], rtol=1e-4, atol=2e-4)
    if 'last_x' in g.files:
        np.testing.assert_allclose(torch.stack([t.detach() for t in lx]).cpu().numpy(), g['last_x'], rtol=1e-4, atol=2e-4)
        assert np.array_equal(torch.stack(cf[:steps - trunc]).cpu().numpy(), g['copy_flag_hard'])
        np.testing.assert_allclose(torch.stack(ct).cpu().numpy(), g['condt'], rtol=1e-6)
        np.testing.assert_allclose(torch.stack(mct).cpu().numpy(), g['mc_t'], rtol=1e-6)
    rw = torch.from_numpy(g['reward_w']).to(dev)
    (-(out.permute(0, 2, 1) * rw).sum((1, 2)).mean()).backward()
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith('grad_backbone'):
            name = key[len('grad_'):]
            ref = torch.from_numpy(g[key])
            got = params[name].grad.cpu()[:ref.shape[0]]
            e = relerr(got, ref)
            print(name, e)
            assert e < 5e-3, (name, e)
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
    assert abs(gn - float(g['grad_norm'])) <= 2e-3 * float(g['grad_norm']), (gn, float(g['grad_norm']))


def test_sample_finetune_gradient_bf16_hard_phase(dev, det_sd):
    """hip_dtype='bf16': the gradient-free hard steps run as one launch of the fused sampler, which records the state each
    step started from; the soft steps and everything differentiable stay fp32."""
    steps, trunc, B = 24, 6, 4
    m = drakes_amd.Diffusion(drakes_amd.default_config(steps=steps, truncate_steps=trunc)).to(dev)
    m.load_state_dict(det_sd)
    m.train()
    ref = m._sample_finetune_gradient(eval_sp_size=B, seed=77)
    m.hip_dtype = 'bf16'
    out, lx, ct, mct, cf = m._sample_finetune_gradient(eval_sp_size=B, seed=77)
    n_hard = steps - trunc
    assert len(lx) == len(ct) == len(mct) == len(cf) == steps and out.requires_grad
    assert float(lx[0][..., 4].min()) == 1.0                              # starts all-masked
    for i in range(n_hard):
        assert lx[i].shape == (B, 200, 5) and cf[i].shape == (B, 200, 1) and cf[i].dtype == torch.int64
        assert torch.equal(cf[i].squeeze(-1), (lx[i].argmax(-1) != 4).long())
        assert torch.equal(ct[i], ref[2][i]) and torch.equal(mct[i], ref[3][i])
        if i:
            keep = cf[i - 1].squeeze(-1).bool()                           # carry-over: an unmasked token never changes
            assert torch.equal(lx[i].argmax(-1)[keep], lx[i - 1].argmax(-1)[keep])
    # same noise, bf16 logits: the recorded trajectory stays close to the fp32 one
    mism = float((lx[n_hard].argmax(-1) != ref[1][n_hard].argmax(-1)).float().mean())
    print('token mismatch vs fp32 hard phase after', n_hard, 'steps:', mism)
    assert mism < 0.05
    (out.sum(-1) * torch.randn(B, 200, device=dev)).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)


@pytest.mark.parametrize('soft', ['fp32', 'bf16'])
def test_flat_parameter_gradients_equal_per_parameter_ones(dev, det_sd, soft):
    """_sample_finetune_gradient hands every soft step ONE flat parameter vector (autograd.flat_params) so that a backward
    call returns one gradient instead of 129; model.hip_flat_grads = False keeps the per-parameter form.  Same numbers
    (up to the order of fp32 atomics), same set of parameters with a gradient, existing .grad accumulated into."""
    grads = {}
    for flat in (False, True):
        m = drakes_amd.Diffusion(drakes_amd.default_config(steps=12, truncate_steps=5)).to(dev)
        m.load_state_dict(det_sd)
        m.train()
        m.hip_soft_dtype = soft
        m.hip_flat_grads = flat
        rw = torch.randn(4, 200, generator=torch.Generator().manual_seed(3)).to(dev)
        for _ in range(2):                                                # two passes: the second accumulates into .grad
            out = m._sample_finetune_gradient(eval_sp_size=3, seed=11)[0]
            (out.permute(0, 2, 1) * rw).sum().backward()
        grads[flat] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    assert grads[False].keys() == grads[True].keys() and len(grads[True]) >= 128
    for k in grads[True]:
        a, b = grads[False][k], grads[True][k]
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-3), k


@pytest.mark.parametrize('B,L,width', [(1, 1, 128), (1, 63, 128), (1, 65, 128), (3, 200, 128), (5, 333, 128), (2, 77, 32)])
def test_ln_bwd_kernel_vs_float64_autograd(dev, B, L, width):
    """drk_ln_bwd alone (LayerNorm backward + residual add + column sums, and the fused hand-over to the layer below),
    against torch autograd in float64 on the same inputs: ragged row counts across its 64-row passes and workgroups, a
    narrow model's statistics width.  Reference: F.layer_norm in models/dnaconv.py:94.  Bar: 2e-5 of each tensor's max
    (fp32 sums in another order)."""
    from drakes_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + L)
    r = lambda *s: torch.randn(*s, generator=g)
    feat, dh, gres = r(B, L, 128), r(B, L, 128), r(B, L, 128)
    tb, gamma = r(128), 1 + 0.1 * r(128)
    if width < 128:
        feat[..., width:] = 0; tb[width:] = 0; gamma[width:] = 0; dh[..., width:] = 0
    mask = (torch.rand(B, L, 128, generator=g) > 0.4).to(torch.uint8)
    x = (feat.double() + tb.double()).requires_grad_()
    gm = gamma.double().requires_grad_()
    h = F.layer_norm(x[..., :width], (width,), gm[:width], None, 1e-5)
    (h * dh.double()[..., :width]).sum().backward()
    dx_ref = torch.zeros(B, L, 128, dtype=torch.float64); dx_ref[..., :width] = x.grad[..., :width]
    dgam_ref = gm.grad
    dbet_ref = dh.double().sum((0, 1))
    o_ref = dx_ref + gres.double()
    xin = (feat + tb)[..., :width]
    mean = xin.mean(-1).reshape(-1)
    rstd = (xin.var(-1, unbiased=False) + 1e-5).rsqrt().reshape(-1)
    D = lambda t: t.to(dev).contiguous()
    dxin, dfeat = torch.empty(B, L, 128, device=dev), torch.empty(B, L, 128, device=dev)
    dgam, dbet, dtb, dbn = (torch.zeros(128, device=dev) for _ in range(4))
    dc = torch.empty(B, L, 128, dtype=torch.bfloat16, device=dev)
    ops.ln_bwd(D(dh), D(feat), D(tb), 0, D(gamma), D(mean), D(rstd), D(gres), dxin, dfeat, dgam, dbet, dtb,
               next_mask=D(mask), dc_next=dc, dbias_next=dbn, width=width)
    bar = lambda got, ref: (got.double().cpu() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1e-3)
    assert bar(dxin, dx_ref) and bar(dfeat, o_ref)
    assert bar(dgam, dgam_ref) and bar(dbet, dbet_ref) and bar(dtb, dx_ref.sum((0, 1)))
    gated = o_ref * mask.double()
    assert bar(dbn, gated.sum((0, 1)))
    assert (dc.float().cpu().double() - gated).abs().max().item() <= 2 ** -8 * max(gated.abs().max().item(), 1e-3)
