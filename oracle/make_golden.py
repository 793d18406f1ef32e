"""TEST INFRASTRUCTURE — generates tests/golden/*.npz from the REAL reference.

Runs only in the build container (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

It imports the reference `Diffusion` class unmodified.  The image lacks lightning,
hydra, torchmetrics, timm and grelu, which the reference imports at module scope but
does not use on this path; minimal stand-ins are registered in sys.modules first
(SURVEY.md section 8c recipe).  Weights come from drakes_amd.det_init (a formula), the
uniform draws from oracle.philox via a patched torch.rand_like, so the fixtures hold
only inputs that cannot be regenerated and the reference's outputs.

Every fixture is also checked here, bit for bit, against oracle/drakes_oracle.py:
that is what pins the oracle.
"""
import os
import sys
import types
from contextlib import contextmanager
from types import SimpleNamespace as NS

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = '/root/reference/drakes_dna'
OUT = os.path.join(ROOT, 'tests', 'golden')


def _install_stand_ins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

        @property
        def dtype(self):
            return torch.float32

    class MeanMetric(nn.Module):
        pass

    class MetricCollection(nn.Module):
        def __init__(self, metrics=None):
            super().__init__()

        def set_dtype(self, dtype):
            pass

        def clone(self, prefix=''):
            return MetricCollection()

    class CosineLRScheduler:
        pass

    lt = mod('lightning', LightningModule=LightningModule)
    lt.pytorch = mod('lightning.pytorch')
    lt.pytorch.utilities = mod('lightning.pytorch.utilities', rank_zero_only=lambda f: f)
    hy = mod('hydra')
    hy.utils = mod('hydra.utils')
    tm = mod('torchmetrics', MetricCollection=MetricCollection)
    tm.aggregation = mod('torchmetrics.aggregation', MeanMetric=MeanMetric)
    ti = mod('timm')
    ti.scheduler = mod('timm.scheduler', CosineLRScheduler=CosineLRScheduler)
    mod('oracle')  # the reference's own oracle.py (gReLU reward model): not on this path


def make_config(length=200, steps=128, time_conditioning=False, hidden_dim=128, stacks=4,
                truncate_steps=3, temp=1.0, eval_batch=8):
    """The ~15 values the hot path reads, as attribute objects."""
    return NS(
        backbone='cnn', parameterization='subs', time_conditioning=time_conditioning, T=0,
        subs_masking=False,
        sampling=NS(predictor='ddpm', steps=steps, noise_removal=True),
        training=NS(antithetic_sampling=True, importance_sampling=False,
                    change_of_variables=False, ema=0.0, sampling_eps=1e-3),
        model=NS(length=length, hidden_dim=hidden_dim, num_cnn_stacks=stacks, dropout=0.0,
                 clean_data=False, cls_free_guidance=False),
        noise=NS(type='loglinear'),
        optim=NS(lr=3e-4),
        loader=NS(eval_batch_size=eval_batch),
        finetuning=NS(gumbel_softmax_temp=temp, truncate_steps=truncate_steps),
    )


@contextmanager
def injected_uniforms(seed):
    """torch.rand_like -> oracle.philox.step_uniforms(seed, call index, ...)."""
    from oracle_pkg import philox
    state = {'step': 0}
    real = torch.rand_like

    def fake(t, *a, **k):
        B, L, V = t.shape
        u = torch.from_numpy(philox.step_uniforms(seed, state['step'], B, L))
        state['step'] += 1
        return u.to(t.dtype)

    torch.rand_like = fake
    try:
        yield state
    finally:
        torch.rand_like = real


def main():
    _install_stand_ins()
    # our oracle package is imported under another name: the reference has its own `oracle` module name
    import importlib.util
    pkg = types.ModuleType('oracle_pkg')
    pkg.__path__ = [HERE]
    sys.modules['oracle_pkg'] = pkg
    for name in ('philox', 'drakes_oracle'):
        spec = importlib.util.spec_from_file_location(f'oracle_pkg.{name}', os.path.join(HERE, f'{name}.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f'oracle_pkg.{name}'] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
    philox, O = pkg.philox, pkg.drakes_oracle

    sys.path.insert(0, REF)
    import diffusion_gosai_update as R
    from drakes_amd.det_init import det_state_dict

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    torch.manual_seed(0)
    WSEED = 7
    sd = det_state_dict(WSEED)

    def ref_model(**kw):
        m = R.Diffusion(make_config(**kw))
        missing = m.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        return m

    def eq(a, b, what):
        assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
        assert torch.equal(a, b), f'oracle != reference for {what}: max diff {(a.double() - b.double()).abs().max()}'

    model = ref_model()
    model_tc = ref_model(time_conditioning=True)

    # ---- F1: per-step scalars --------------------------------------------------------
    rows = []
    ts = torch.linspace(1, 1e-5, 129)
    dt = (1 - 1e-5) / 128
    for i in range(128):
        t = ts[i] * torch.ones(1, 1)
        s_t, _ = model.noise(t)
        s_s, _ = model.noise(t - dt)
        rows.append(torch.stack([t[0, 0], s_t[0, 0], 1 - torch.exp(-s_t[0, 0]), 1 - torch.exp(-s_s[0, 0])]))
    tab = torch.stack(rows)
    eq(O.step_scalars(128), tab, 'step scalars')
    sig_last = model.noise(ts[-1] * torch.ones(1, 1))[0][0, 0]
    np.savez_compressed(os.path.join(OUT, 'f1_step_scalars.npz'), table_bits=tab.numpy().view(np.uint32),
             sigma_last_bits=sig_last.numpy().reshape(1).view(np.uint32))

    # ---- F2: backbone logits, int and soft input, with and without time conditioning --
    g = np.random.Generator(np.random.PCG64(11))
    B, L = 4, 200
    x_int = torch.from_numpy(g.integers(0, 5, (B, L)))
    x_int[0] = 4                      # all masked
    x_int[1, :] = torch.from_numpy(g.integers(0, 4, L))   # none masked
    soft = torch.from_numpy(g.dirichlet(np.ones(5) * 0.3, (B, L)).astype(np.float32))
    sigma = torch.tensor([0.0078452928, 0.69215769, 3.0, 6.9077682], dtype=torch.float32)
    with torch.no_grad():
        lg_int = model.backbone(x_int, torch.zeros(B))
        lg_soft = model.backbone(soft, torch.zeros(B))
        lg_int_tc = model_tc.backbone(x_int, sigma)
        lg_soft_tc = model_tc.backbone(soft, sigma)
        eq(O.cnn_forward(sd, x_int, torch.zeros(B)), lg_int, 'logits int')
        eq(O.cnn_forward(sd, soft, torch.zeros(B)), lg_soft, 'logits soft')
        eq(O.cnn_forward(sd, x_int, sigma), lg_int_tc, 'logits int tc')
        eq(O.cnn_forward(sd, soft, sigma), lg_soft_tc, 'logits soft tc')
        temb = torch.relu(model.backbone.time_embedder(sigma))
        eq(O.time_embedding(sd, sigma), temb, 'time embedding')
        # Diffusion.forward (SUBS on top), both input kinds
        lp_int = model.forward(x_int, sigma[:, None])
        lp_soft = model.forward(soft, sigma[:, None])
        lp_int_tc = model_tc.forward(x_int, sigma[:, None])
        eq(O.model_forward(sd, x_int, sigma[:, None]), lp_int, 'forward int')
        eq(O.model_forward(sd, soft, sigma[:, None]), lp_soft, 'forward soft')
        eq(O.model_forward(sd, x_int, sigma[:, None], True), lp_int_tc, 'forward int tc')
    np.savez_compressed(os.path.join(OUT, 'f2_backbone.npz'), wseed=WSEED, x_int=x_int.numpy(), soft=soft.numpy(),
             sigma=sigma.numpy(), logits_int=lg_int.numpy(), logits_soft=lg_soft.numpy(),
             logits_int_tc=lg_int_tc.numpy(), logits_soft_tc=lg_soft_tc.numpy(), time_emb=temb.numpy(),
             logp_int=lp_int.numpy(), logp_soft=lp_soft.numpy(), logp_int_tc=lp_int_tc.numpy())

    # ---- F3: hard reverse step at several steps / masking patterns --------------------
    # The backbone is replaced by a table lookup so that the step arithmetic is pinned on
    # its own: (logits, x, U, mc_t, mc_s) -> x_next through the reference's _ddpm_update.
    f3 = {}
    B3, L3 = 6, 200
    USEED = 2024
    for k, step in enumerate((0, 1, 64, 126, 127)):
        logits = torch.from_numpy((g.standard_normal((B3, L3, 5)) * 2.0).astype(np.float32))
        x = torch.from_numpy(g.integers(0, 5, (B3, L3)))
        x[0] = 4
        x[1] = torch.from_numpy(g.integers(0, 4, L3))
        x[2, ::2] = 4
        real_backbone = model.backbone.forward
        model.backbone.forward = lambda seq, t, _l=logits: _l.clone()
        t = ts[step] * torch.ones(B3, 1)
        with injected_uniforms(USEED) as st, torch.no_grad():
            st['step'] = step
            x_next = model._ddpm_update(x, t, dt)
        model.backbone.forward = real_backbone
        u = torch.from_numpy(philox.step_uniforms(USEED, step, B3, L3))
        mine = O.hard_step_from_logits(logits.clone(), x, u, tab[step, 2], tab[step, 3])
        eq(mine, x_next, f'hard step {step}')
        f3[f'logits_{k}'] = logits.numpy()
        f3[f'x_{k}'] = x.numpy()
        f3[f'xnext_{k}'] = x_next.numpy()
        f3[f'step_{k}'] = step
    np.savez_compressed(os.path.join(OUT, 'f3_hard_step.npz'), useed=USEED, n=5, **f3)

    # ---- F4: _sample tokens with injected uniforms ------------------------------------
    for tag, m, tc in (('', model, False), ('_tc', model_tc, True)):
        B4 = 8 if not tc else 4
        with injected_uniforms(4242):
            toks = m._sample(eval_sp_size=B4)
        trace = []
        mine = O.sample(sd, B4, 200, 128, lambda i, shp: torch.from_numpy(philox.step_uniforms(4242, i, shp[0], shp[1])),
                        time_conditioning=tc, trace=trace)
        eq(mine, toks, f'_sample tokens{tag}')
        np.savez_compressed(os.path.join(OUT, f'f4_sample{tag}.npz'), wseed=WSEED, useed=4242, tokens=toks.numpy(),
                 state_after_32=trace[31].numpy(), state_after_96=trace[95].numpy(), state_after_128=trace[127].numpy())

    # short-schedule variant (other num_steps / length) for cheap GPU tests
    m_short = ref_model(length=64, steps=16)
    with injected_uniforms(99):
        toks = m_short._sample(eval_sp_size=5)
    mine = O.sample(sd, 5, 64, 16, lambda i, shp: torch.from_numpy(philox.step_uniforms(99, i, shp[0], shp[1])))
    eq(mine, toks, '_sample short')
    np.savez_compressed(os.path.join(OUT, 'f4_sample_short.npz'), wseed=WSEED, useed=99, tokens=toks.numpy())

    # ---- F5: _sample_finetune_gradient -----------------------------------------------
    names = ['backbone.final_conv.2.weight', 'backbone.linear.weight', 'backbone.norms.19.weight',
             'backbone.time_layers.0.dense.bias', 'backbone.convs.7.weight', 'backbone.convs.19.bias']
    rw = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal((4, 200)).astype(np.float32))

    def run_f5(tag, steps, trunc, B5, temp, cft, full_lists):
        m = ref_model(steps=steps, truncate_steps=trunc, temp=temp)
        m.train()
        with injected_uniforms(777):
            out, lx, ct, mct, cf = m._sample_finetune_gradient(eval_sp_size=B5, copy_flag_temp=cft)
        reward = (out.permute(0, 2, 1) * rw).sum((1, 2))       # synthetic linear reward on [B,4,L]
        (-reward.mean()).backward()
        grads = {n: dict(m.named_parameters())[n].grad.clone() for n in names}
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)).float()

        sd_g = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in sd.items()}
        o2, lx2, ct2, mct2, cf2 = O.sample_finetune_gradient(
            sd_g, B5, 200, steps, trunc,
            lambda i, shp: torch.from_numpy(philox.step_uniforms(777, i, shp[0], shp[1])),
            temp=temp, copy_flag_temp=cft)
        eq(o2, out.detach(), f'finetune sample {tag}')
        for a, b in zip(lx2, lx):
            eq(a.detach(), b.detach(), 'last_x')
        for a, b in zip(cf2, cf):
            eq(a.detach(), b.detach(), 'copy_flag')
        for a, b in zip(ct2, ct):
            eq(a, b, 'condt')
        for a, b in zip(mct2, mct):
            eq(a, b, 'mc_t')
        r2 = (o2.permute(0, 2, 1) * rw).sum((1, 2))
        (-r2.mean()).backward()
        for n in names:
            d = (sd_g[n].grad - grads[n]).abs().max().item()
            s = grads[n].abs().max().item()
            assert d <= 1e-5 * max(s, 1e-3), (n, d, s)
        save = dict(wseed=WSEED, useed=777, steps=steps, trunc=trunc, temp=temp,
                    cft=(-1.0 if cft is None else cft), sample=out.detach().numpy(),
                    reward_w=rw.numpy(), grad_norm=gn.numpy(),
                    last_x_final=lx[-1].detach().numpy(), copy_flag_final=cf[-1].detach().numpy())
        for n in names:
            # the [128,128,9] conv gradient is 590 KB: keep its first 16 output channels
            save['grad_' + n] = grads[n].numpy()[:16] if n.endswith('convs.7.weight') else grads[n].numpy()
        if full_lists:
            save['last_x'] = torch.stack([t.detach() for t in lx]).numpy()
            save['copy_flag_soft'] = torch.stack([t.detach() for t in cf[steps - trunc:]]).numpy()
            save['copy_flag_hard'] = torch.stack([t for t in cf[:steps - trunc]]).numpy()
            save['condt'] = torch.stack(ct).numpy()
            save['mc_t'] = torch.stack(mct).numpy()
        np.savez_compressed(os.path.join(OUT, f'f5_finetune_{tag}.npz'), **save)

    run_f5('s16k6', 16, 6, 2, 1.0, None, True)
    run_f5('s16k6_cft', 16, 6, 2, 0.7, 0.5, False)
    run_f5('s128k50', 128, 50, 2, 1.0, None, False)
    print('golden fixtures written to', OUT)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
