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


if __name__ == '__main__':
    main()
