"""One rank of the data-parallel tests (started by tests/test_dp.py and tests/test_gpu_dp.py, one process per
rank, rendezvous on 127.0.0.1 from the environment torchrun would set).

mode cpu: drakes_amd.dp driven with the CPU oracle standing in for the HIP sampler (the test is about the sharding,
          the bucket and the collective, over gloo);
mode gpu: the real HIP path on cuda:0 in every rank, gloo collectives on device tensors (two RCCL ranks cannot
          share one GPU, and gpurun boxes have one).
Writes rank{r}.pt into the output directory.
"""
import argparse
import faulthandler
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import drakes_amd                                    # noqa: E402
from drakes_amd import dp                            # noqa: E402
from drakes_amd.det_init import det_state_dict       # noqa: E402


def reward_weights(global_batch, L):
    g = torch.Generator().manual_seed(11)
    return torch.randn(global_batch, 4, L, generator=g)


class OracleBacked:
    """`_sample` / `_sample_finetune_gradient` of the CPU oracle behind the method names dp.py calls."""

    def __init__(self, model, L):
        from oracle import drakes_oracle as O, philox
        self.O, self.philox, self.model, self.L = O, philox, model, L

    def _u(self, seed, count, first):
        return lambda i, shape: torch.from_numpy(self.philox.step_uniforms(seed, i, count, self.L, first))

    def _sample(self, num_steps=None, eps=1e-5, eval_sp_size=None, *, seed, sample_offset=0):
        sd = {k: v.detach() for k, v in self.model.state_dict(keep_vars=True).items()}
        return self.O.sample(sd, eval_sp_size, self.L, num_steps, self._u(seed, eval_sp_size, sample_offset), eps)

    def _sample_finetune_gradient(self, num_steps, eval_sp_size, trunc, *, seed, sample_offset=0):
        sd = dict(self.model.state_dict(keep_vars=True))
        return self.O.sample_finetune_gradient(sd, eval_sp_size, self.L, num_steps, trunc,
                                               self._u(seed, eval_sp_size, sample_offset))[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', choices=['cpu', 'gpu'], required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--global-batch', type=int, default=5)
    ap.add_argument('--grad-batch', type=int, default=4)
    ap.add_argument('--length', type=int, default=40)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--trunc', type=int, default=2)
    ap.add_argument('--stacks', type=int, default=1)
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--hip-dtype', default='fp32')
    ap.add_argument('--finetune', action='store_true', help='also run one drakes_amd.fine_tune optimizer step (gpu mode)')
    a = ap.parse_args()
    faulthandler.dump_traceback_later(150, exit=True)      # a stuck collective names itself instead of hanging

    print('worker start', os.environ.get('RANK'), flush=True)
    rank, world, device = dp.init(backend='gloo', device='cuda:0' if a.mode == 'gpu' else 'cpu')
    cfg = drakes_amd.default_config(length=a.length, steps=a.steps, num_cnn_stacks=a.stacks, truncate_steps=a.trunc)
    model = drakes_amd.Diffusion(cfg).to(device)
    model.load_state_dict(det_state_dict(7, 128, a.stacks))
    print('group up', rank, world, flush=True)
    sampler = model if a.mode == 'gpu' else OracleBacked(model, a.length)
    if a.mode == 'gpu':
        model.hip_dtype = a.hip_dtype

    # (1) sharded sampling, uneven shards allowed; every rank ends with all rows
    tokens, first = dp.sample_sharded(sampler, a.global_batch, seed=a.seed, num_steps=a.steps, gather=True)
    mine, mine_first = dp.sample_sharded(sampler, a.global_batch, seed=a.seed, num_steps=a.steps)

    print('sampled', flush=True)
    # (2) one fine-tune gradient exchange: loss = mean over this rank's rows; equal shards
    bucket = dp.GradBucket(model)
    bucket.zero_()
    gfirst, gcount = dp.shard_range(a.grad_batch, rank, world)
    rw = reward_weights(a.grad_batch, a.length)[gfirst:gfirst + gcount].to(device)
    model.train()
    if a.mode == 'gpu':
        out = model._sample_finetune_gradient(eval_sp_size=gcount, seed=a.seed, sample_offset=gfirst)[0]
    else:
        out = sampler._sample_finetune_gradient(a.steps, gcount, a.trunc, seed=a.seed, sample_offset=gfirst)
    loss = -(out.permute(0, 2, 1) * rw).sum((1, 2)).mean()
    loss.backward()
    print('backward done', flush=True)
    local_flat = bucket.flat.clone()
    bucket.allreduce_mean_()
    norm = bucket.clip_grad_norm_(1.0)
    views_ok = all(p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * off for p, off in zip(bucket.params, bucket.offsets))

    ft = {}
    if a.finetune and a.mode == 'gpu':
        from types import SimpleNamespace as NS
        from oracle.finetune_oracle import SyntheticReward
        model.hip_dtype = 'fp32'
        start = {k: v.detach().clone() for k, v in model.state_dict().items()}
        old_model = drakes_amd.Diffusion(cfg).to(device)
        old_model.load_state_dict(det_state_dict(8, 128, a.stacks))
        fargs = NS(truncate_steps=a.trunc, gumbel_temp=1.0, total_num_steps=a.steps, learning_rate=1e-4, num_epochs=1,
                   num_accum_steps=2, batch_size=a.grad_batch // world, copy_flag_temp=None, truncate_kl=False, alpha=0.01,
                   alpha_schedule_warmup=0, gradnorm_clip=1.0, save_every_n_epochs=10 ** 9)
        reward = SyntheticReward(3).to(device)
        losses = drakes_amd.fine_tune(model, reward, reward, old_model, fargs, seed_of_call=lambda i: 500 + i, verbose=False)
        ft = {'ft_losses': torch.tensor([float(x) for x in losses]),
              'ft_delta': torch.cat([(model.state_dict()[k] - start[k]).flatten().cpu()
                                     for k in ('backbone.final_conv.2.weight', 'backbone.norms.3.weight', 'backbone.linear.weight')]),
              'ft_norm': drakes_amd.fine_tune.last_log['tot_grad_norm']}

    torch.save({**ft, 'tokens': tokens.cpu(), 'first': first, 'mine': mine.cpu(), 'mine_first': mine_first,
                'flat': bucket.flat.cpu(), 'local_flat': local_flat.cpu(), 'norm': float(norm),
                'views_ok': views_ok, 'loss': float(loss)},
               os.path.join(a.out, f'rank{rank}.pt'))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
