"""TEST INFRASTRUCTURE — fixture F8: the reverse loop of the REAL protein sampler, `Interpolant.sample`
(/root/reference/drakes_protein/fmif/fm_utils.py:86-139), with a table in place of ProteinMPNN (the denoiser is outside
the path) and torch.rand_like patched to oracle.philox.step_uniforms_wide.  Checks oracle.drakes_oracle.hard_step_wide.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_protein.py        (build container only)
"""
import os
import sys
import types
from types import SimpleNamespace as NS

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
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
    sys.path.insert(0, '/root/reference/drakes_protein')
    from fmif import fm_utils
    from fmif import model_utils as mu
    assert mu.MASK_TOKEN_INDEX == 21

    B, L, V, T, SEED = 3, 57, 22, 9, 606
    g = np.random.Generator(np.random.PCG64(23))
    table = torch.from_numpy((g.standard_normal((T - 1, B, L, V)) * 2.5).astype(np.float32))
    cfg = NS(num_timesteps=T, min_t=1e-2, temp=0.7, interpolant_type='masking')
    it = fm_utils.Interpolant(cfg)
    it.set_device(torch.device('cpu'))
    state = {'k': 0}

    def model(X, S, mask, chain_M, residue_idx, chain_encoding_all, cls=None):
        out = table[state['k']].clone()
        return out
    real = torch.rand_like

    def fake(t, *a, **k):
        u = torch.from_numpy(philox.step_uniforms_wide(SEED, state['k'], B, L, V))
        state['k'] += 1
        return u
    torch.rand_like = fake
    try:
        mask = torch.ones(B, L)
        pred, traj, clean = it.sample(model, None, mask, mask, None, None)
    finally:
        torch.rand_like = real
    # the oracle step, chained
    ts = torch.linspace(cfg.min_t, 1.0, T)
    x = torch.full((B, L), 21, dtype=torch.int64)
    t_1 = ts[0]
    for k, t_2 in enumerate(ts[1:]):
        u = torch.from_numpy(philox.step_uniforms_wide(SEED, k, B, L, V))
        x, pc = O.hard_step_wide(table[k], x, u, t_2 - t_1, 1.0 - t_2, temp=cfg.temp)
        assert torch.equal(x, traj[k + 1].long()), k
        assert torch.equal(pc, clean[k]), k
        t_1 = t_2
    assert torch.equal(pc, pred)
    print('masked left per step:', [(t == 21).sum().item() for t in traj])
    np.savez_compressed(os.path.join(OUT, 'f8_protein_sample.npz'), useed=SEED, tseed=23, T=T, min_t=cfg.min_t, temp=cfg.temp,
                        table=table.numpy(), traj=torch.stack([t.long() for t in traj]).numpy(),
                        clean=torch.stack(clean).numpy())
    print('written', os.path.getsize(os.path.join(OUT, 'f8_protein_sample.npz')))

    # ---- F9: sample_gradient (fm_utils.py:142-209): hard steps, then Gumbel-softmax steps with gradients -------------
    import torch.nn.functional as F
    T9, K9, GT = 9, 5, 0.6
    g9 = np.random.Generator(np.random.PCG64(29))
    tab = torch.from_numpy((g9.standard_normal((T9 - 1, B, L, V)) * 2.0).astype(np.float32))
    mix = torch.from_numpy((g9.standard_normal((V, V)) * 0.3).astype(np.float32))      # makes the "model" depend on its input
    W = torch.from_numpy(g9.standard_normal((B, L, V)).astype(np.float32))
    cfg9 = NS(num_timesteps=T9, min_t=1e-2, temp=0.8, interpolant_type='masking')

    def run_reference():
        t = tab.clone().requires_grad_()
        it9 = fm_utils.Interpolant(cfg9)
        it9.set_device(torch.device('cpu'))
        st = {'k': 0, 'm': 0}

        def model9(X, S, mask, chain_M, residue_idx, chain_encoding_all, cls=None):
            out = t[st['m']] + S @ mix
            st['m'] += 1
            return out

        def fake9(tt, *a, **k):
            u = torch.from_numpy(philox.step_uniforms_wide(SEED + 1, st['k'], B, L, V))
            st['k'] += 1
            return u
        torch.rand_like = fake9
        try:
            m1 = torch.ones(B, L)
            res = it9.sample_gradient(model9, None, m1, m1, None, None, K9, GT)
        finally:
            torch.rand_like = real
        (res[0] * W).sum().backward()
        return res, t.grad

    (out9, lx9, mct9, cf9), grad9 = run_reference()
    t2 = tab.clone().requires_grad_()
    o2, lx2, mct2, cf2 = O.sample_gradient_wide(lambda k, xs: t2[k] + xs @ mix, B, L, T9, cfg9.min_t, cfg9.temp, K9, GT,
                                                lambda k: torch.from_numpy(philox.step_uniforms_wide(SEED + 1, k, B, L, V)))
    (o2 * W).sum().backward()
    assert torch.equal(o2.detach(), out9.detach())
    for a, b in zip(lx2, lx9):
        assert torch.equal(a.detach(), b.detach())
    for a, b in zip(cf2, cf9):
        assert torch.equal(a.detach(), b.detach())
    assert all(float(a) == float(b) for a, b in zip(mct2, mct9))
    assert float((t2.grad - grad9).abs().max()) <= 1e-6 * float(grad9.abs().max())
    print('soft steps with gradient:', [k for k in range(T9 - 1) if float(grad9[k].abs().max()) > 0])
    np.savez_compressed(os.path.join(OUT, 'f9_protein_sample_gradient.npz'), useed=SEED + 1, T=T9, K=K9, gumbel_temp=GT,
                        min_t=cfg9.min_t, temp=cfg9.temp, table=tab.numpy(), mix=mix.numpy(), W=W.numpy(),
                        sample=out9.detach().numpy(), last_x_final=lx9[-1].detach().numpy(),
                        last_x_first_soft=lx9[T9 - K9 + 1].detach().numpy(), copy_flag_final=cf9[-1].detach().numpy(),
                        mct=np.array([float(m) for m in mct9], dtype=np.float32), grad_table=grad9.numpy())
    print('written', os.path.getsize(os.path.join(OUT, 'f9_protein_sample_gradient.npz')))


if __name__ == '__main__':
    main()
