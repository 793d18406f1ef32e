"""TEST INFRASTRUCTURE — generates tests/golden/f6_finetune_loop_*.npz from the REAL reference fine-tune loop.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_finetune.py        (build container only)

/root/reference/drakes_dna/finetune_reward_bp.py is a script: importing it would parse the command line, start
Hydra and load checkpoints this image does not have.  Its `fine_tune` function (:17-123) is therefore taken out of the
parsed source (ast) and executed UNMODIFIED in a namespace that supplies the module-level names it reads (torch, F,
np, os, log_path, save_path, wandb).  The model is the real reference `Diffusion` (stand-ins as in make_golden.py), the
reward model is oracle.finetune_oracle.SyntheticReward (the gReLU model is absent), the uniform draws come from
oracle.philox through a patched torch.rand_like, one Philox seed per `_sample_finetune_gradient` call.
oracle/finetune_oracle.py is checked against it here (losses and updated weights), which pins it.
"""
import ast
import os
import sys
import tempfile
import types
from types import SimpleNamespace as NS

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

REF = MG.REF
OUT = MG.OUT
BASE_SEED = 9000
NAMES = ['backbone.final_conv.2.weight', 'backbone.linear.weight', 'backbone.norms.19.weight',
         'backbone.time_layers.0.dense.bias', 'backbone.convs.7.weight', 'backbone.convs.19.bias']


def reference_fine_tune():
    src = open(os.path.join(REF, 'finetune_reward_bp.py')).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'fine_tune')
    mod = ast.Module(body=[fn], type_ignores=[])
    tmp = tempfile.mkdtemp()
    ns = {'torch': torch, 'F': F, 'np': np, 'os': os, 'wandb': None, 'log_path': os.path.join(tmp, 'log.txt'),
          'save_path': tmp}
    exec(compile(mod, 'finetune_reward_bp.py', 'exec'), ns)
    return ns['fine_tune']


def main():
    MG._install_stand_ins()
    import importlib.util
    pkg = types.ModuleType('oracle_pkg')
    pkg.__path__ = [HERE]
    sys.modules['oracle_pkg'] = pkg
    for name in ('philox', 'drakes_oracle', 'finetune_oracle'):
        spec = importlib.util.spec_from_file_location(f'oracle_pkg.{name}', os.path.join(HERE, f'{name}.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f'oracle_pkg.{name}'] = m
        spec.loader.exec_module(m)
        setattr(pkg, name, m)
    philox, FO = pkg.philox, pkg.finetune_oracle

    sys.path.insert(0, REF)
    import diffusion_gosai_update as R
    from drakes_amd.det_init import det_state_dict

    torch.set_num_threads(8)
    fine_tune = reference_fine_tune()
    WSEED = 7

    def run(tag, steps, trunc, B, epochs, accum, alpha, warmup, truncate_kl, temp, cft, lr, stacks=4):
        sd = det_state_dict(WSEED, 128, stacks)
        g = np.random.Generator(np.random.PCG64(21))
        sd_old = {k: (v + 0.02 * torch.from_numpy(g.standard_normal(tuple(v.shape)).astype(np.float32))
                      if k != 'backbone.time_embedder.0.W' else v.clone()) for k, v in sd.items()}
        args = NS(truncate_steps=trunc, gumbel_temp=temp, total_num_steps=steps, learning_rate=lr, num_epochs=epochs,
                  num_accum_steps=accum, batch_size=B, copy_flag_temp=cft, truncate_kl=truncate_kl, alpha=alpha,
                  alpha_schedule_warmup=warmup, gradnorm_clip=1.0, name='debug', save_every_n_epochs=10 ** 9)

        def ref_model(state):
            m = R.Diffusion(MG.make_config(steps=steps, stacks=stacks))
            m.load_state_dict(state, strict=True)
            return m
        new_model, old_model = ref_model(sd), ref_model(sd_old)
        assert list(new_model.state_dict().keys()) == list(sd.keys())
        reward, reward_eval = FO.SyntheticReward(3), FO.SyntheticReward(4)

        state = {'step': 0, 'seed': BASE_SEED, 'call': 0}
        orig = new_model._sample_finetune_gradient

        def sampling_call(*a, **k):
            state['step'], state['seed'] = 0, BASE_SEED + state['call']
            state['call'] += 1
            return orig(*a, **k)
        new_model._sample_finetune_gradient = sampling_call
        real = torch.rand_like

        def fake(t, *a, **k):
            Bt, Lt, _ = t.shape
            u = torch.from_numpy(philox.step_uniforms(state['seed'], state['step'], Bt, Lt))
            state['step'] += 1
            return u.to(t.dtype)
        torch.rand_like = fake
        try:
            losses = fine_tune(new_model, reward, reward_eval, old_model, args)
        finally:
            torch.rand_like = real
        torch.set_grad_enabled(True)
        losses = np.array([float(x) for x in losses], dtype=np.float32)
        final = {k: v.detach().clone() for k, v in new_model.state_dict().items()}

        # ---- the oracle restatement on the same inputs
        sd_new = {k: v.clone().requires_grad_(k != 'backbone.time_embedder.0.W') for k, v in sd.items()}
        pnames = [k for k in sd if k != 'backbone.time_embedder.0.W']
        first = {}
        l2, log = FO.fine_tune(sd_new, sd_old, pnames, reward, reward_eval, args, 200, lambda i: BASE_SEED + i,
                               record_first_grads=first)
        l2 = np.array([float(x) for x in l2], dtype=np.float32)
        print(tag, 'reference losses', losses, 'oracle', l2)
        # Same ops, but autograd sums the 2*steps+K gradient contributions of a weight in graph order, which differs
        # between the reference's module graph and the functional restatement: equal to fp32 rounding, not bitwise.
        assert np.array_equal(losses[:accum], l2[:accum]), 'oracle fine_tune losses != reference before the first update'
        assert np.allclose(losses, l2, rtol=2e-5, atol=1e-6), 'oracle fine_tune losses != reference'
        worst = max(float((final[k] - sd_new[k].detach()).abs().max()) for k in sd)
        print(tag, 'max |weight(oracle) - weight(reference)| =', worst, '=', worst / lr, 'x lr')
        assert worst <= 0.02 * lr, worst
        save = dict(wseed=WSEED, base_seed=BASE_SEED, steps=steps, trunc=trunc, batch=B, epochs=epochs, accum=accum,
                    alpha=alpha, warmup=warmup, truncate_kl=int(truncate_kl), temp=temp, cft=(-1.0 if cft is None else cft),
                    lr=lr, stacks=stacks, old_perturb_seed=21, losses=losses,
                    grad_norms=np.array([e['grad_norm'] for e in log], dtype=np.float32),
                    kl_losses=np.array([e['kl_loss'] for e in log], dtype=np.float32),
                    mean_rewards=np.array([e['mean_reward'] for e in log], dtype=np.float32),
                    param_sum=np.array([float(sum(v.double().sum() for v in final.values()))]))
        for n in NAMES:
            if n in final:
                d = (final[n] - sd[n]).numpy()                      # the UPDATE (Adam steps), not the weight
                save['delta_' + n] = d[:16] if n.endswith('convs.7.weight') else d
                # accumulated gradient at the first optimizer step, before clipping: from the oracle run, whose weights
                # after every step were just shown to be bit-equal to the reference's
                gr = first[n].numpy()
                save['grad1_' + n] = gr[:16] if n.endswith('convs.7.weight') else gr
        np.savez_compressed(os.path.join(OUT, f'f6_finetune_loop_{tag}.npz'), **save)

    # ONE optimizer step (tight checks: same weights in, so gradients and the Adam update are comparable entry by entry)
    run('a', steps=12, trunc=4, B=2, epochs=1, accum=2, alpha=0.01, warmup=0, truncate_kl=False, temp=1.0, cft=None, lr=1e-4)
    run('b', steps=12, trunc=5, B=2, epochs=1, accum=1, alpha=0.05, warmup=2, truncate_kl=True, temp=0.8, cft=0.5, lr=1e-4)
    # three optimizer steps (loose checks: later steps see tokens sampled by argmax from already-updated weights)
    run('c', steps=12, trunc=5, B=2, epochs=3, accum=1, alpha=0.05, warmup=2, truncate_kl=True, temp=0.8, cft=0.5, lr=1e-4)
    for f in sorted(os.listdir(OUT)):
        if f.startswith('f6'):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
