"""TEST INFRASTRUCTURE — fixture F7: Diffusion.get_likelihood / q_xt of the REAL reference
(/root/reference/drakes_dna/diffusion_gosai_update.py:425-436, 885-926) with torch.rand patched to the first Philox
uniform of each position; checks oracle.drakes_oracle.get_likelihood against it.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_likelihood.py        (build container only)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    MG._install_stand_ins()
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
    sys.path.insert(0, MG.REF)
    import diffusion_gosai_update as R
    from drakes_amd.det_init import det_state_dict

    torch.set_num_threads(8)
    sd = det_state_dict(7)
    SEED, B, L, NS = 31337, 4, 200, 2
    g = np.random.Generator(np.random.PCG64(17))
    x0 = torch.from_numpy(g.integers(0, 4, (B, L)))
    out = {}
    for tag, tc, steps in (('', False, 128), ('_tc', True, 32)):
        model = R.Diffusion(MG.make_config(steps=steps, time_conditioning=tc))
        model.load_state_dict(sd, strict=True)
        state = {'call': 0}
        real = torch.rand

        def fake(*shape, **kw):
            s, i = divmod(state['call'], steps)
            state['call'] += 1
            assert tuple(shape) == (B, L)
            return torch.from_numpy(philox.step_uniforms(SEED + s, i, B, L)[:, :, 0].copy())
        torch.rand = fake
        try:
            ll = model.get_likelihood(x0, n_samples=NS)
        finally:
            torch.rand = real
        mine = O.get_likelihood(sd, x0, steps, lambda s, i: torch.from_numpy(philox.step_uniforms(SEED + s, i, B, L)[:, :, 0].copy()),
                                n_samples=NS, time_conditioning=tc)
        assert torch.equal(mine, ll), (mine, ll)
        print(tag or 'default', ll)
        out['loglik' + tag] = ll.numpy()
        out['steps' + tag] = steps
    # q_xt alone
    mc = 0.37
    u = torch.from_numpy(philox.step_uniforms(SEED, 5, B, L)[:, :, 0].copy())
    torch.rand = lambda *a, **k: u
    try:
        xt = model.q_xt(x0, torch.full((B, 1), mc))
    finally:
        torch.rand = real
    assert torch.equal(O.q_xt(x0, mc, u), xt)
    np.savez_compressed(os.path.join(MG.OUT, 'f7_likelihood.npz'), wseed=7, useed=SEED, x0=x0.numpy(), n_samples=NS,
                        qxt_step=5, qxt_mc=mc, qxt=xt.numpy(), **out)
    print('written', os.path.getsize(os.path.join(MG.OUT, 'f7_likelihood.npz')))


if __name__ == '__main__':
    main()
