"""TEST INFRASTRUCTURE — CPU restatement of the DRAKES fine-tune loop (the direct caller of the sampling hot path):
`fine_tune` of /root/reference/drakes_dna/finetune_reward_bp.py:17-123, on the functional oracle of
oracle/drakes_oracle.py.  Only tests/ may import this.  Pinned by oracle/make_golden_finetune.py, which runs the
reference's own `fine_tune` function (extracted from its source file and executed unmodified) on the same weights,
reward model and injected uniforms, asserts equality and writes tests/golden/f6_*.npz.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

try:                                   # imported as oracle.finetune_oracle (tests) or oracle_pkg.* (make_golden)
    from . import drakes_oracle as O
    from . import philox
except ImportError:                    # pragma: no cover
    import drakes_oracle as O
    import philox


class SyntheticReward(nn.Module):
    """Stand-in for the gReLU enhancer-activity model (reference oracle.py, absent here: parity unpinned at that
    boundary).  Same call shape: [B, 4, L] float -> [B, 3, 1]; deterministic weights."""

    def __init__(self, seed=3):
        super().__init__()
        g = np.random.Generator(np.random.PCG64(seed))
        self.conv = nn.Conv1d(4, 8, kernel_size=5, padding=2)
        self.lin = nn.Linear(8, 3)
        with torch.no_grad():
            self.conv.weight.copy_(torch.from_numpy((g.standard_normal((8, 4, 5)) * 0.4).astype(np.float32)))
            self.conv.bias.copy_(torch.from_numpy((g.standard_normal(8) * 0.1).astype(np.float32)))
            self.lin.weight.copy_(torch.from_numpy((g.standard_normal((3, 8)) * 0.5).astype(np.float32)))
            self.lin.bias.zero_()
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        return self.lin(torch.tanh(self.conv(x)).mean(-1)).unsqueeze(-1)


def fine_tune(sd_new, sd_old, param_names, reward, reward_eval, args, length, seed_of_call, time_conditioning=False,
              eps=1e-5, record_first_grads=None):
    """finetune_reward_bp.py:17-123 without logging/checkpoint I/O.

    sd_new: dict name -> leaf tensor (requires_grad on every name in param_names); updated in place by Adam.
    seed_of_call(i): Philox seed of the i-th _sample_finetune_gradient call.  Returns (batch_losses, log).
    record_first_grads: optional dict, filled with name -> accumulated gradient at the first optimizer step (before
    clipping)."""
    params = [sd_new[n] for n in param_names]
    optim = torch.optim.Adam(params, lr=args.learning_rate)                                   # :27
    batch_losses, log = [], []
    call = 0
    for epoch_num in range(args.num_epochs):                                                  # :31
        rewards, losses, reward_losses, kl_losses = [], [], [], []
        tot_grad_norm = 0.0
        for _step in range(args.num_accum_steps):                                             # :39
            seed = seed_of_call(call)
            call += 1
            uni = lambda i, shp, _s=seed: torch.from_numpy(philox.step_uniforms(_s, i, shp[0], shp[1]))
            sample, last_x_list, condt_list, mct_list, flag_list = O.sample_finetune_gradient(
                sd_new, args.batch_size, length, args.total_num_steps, args.truncate_steps, uni, eps=eps,
                temp=args.gumbel_temp, copy_flag_temp=args.copy_flag_temp, time_conditioning=time_conditioning)   # :40
            sample2 = torch.transpose(sample, 1, 2)
            preds = reward(sample2).squeeze(-1)
            rew = preds[:, 0]                                                                 # :41-43
            sample_argmax = 1.0 * F.one_hot(torch.argmax(sample, 2), num_classes=4)
            sample_argmax = torch.transpose(sample_argmax, 1, 2)
            rewards.append(reward(sample_argmax).squeeze(-1)[:, 0].detach().numpy())          # :45-51
            total_kl = []
            for t in range(args.total_num_steps):                                             # :60-75
                if args.truncate_kl and t < args.total_num_steps - args.truncate_steps:
                    continue
                last_x, condt, mct, flag = last_x_list[t], condt_list[t], mct_list[t], flag_list[t]
                lp = O.model_forward(sd_new, last_x, condt, time_conditioning)[:, :, :-1]
                lp_old = O.model_forward(sd_old, last_x, condt, time_conditioning)[:, :, :-1]
                p, p_old = lp.exp(), lp_old.exp()
                kl = flag * (-p + p_old + p * (lp - lp_old)) / mct[0, 0, 0]
                total_kl.append((kl * last_x[:, :, :-1]).sum((1, 2)))
            if epoch_num < args.alpha_schedule_warmup:                                        # :77-81
                alpha = (epoch_num + 1) / args.alpha_schedule_warmup * args.alpha
            else:
                alpha = args.alpha
            kl_loss = torch.stack(total_kl, 1).sum(1).mean()                                  # :83
            reward_loss = -torch.mean(rew)
            loss = (reward_loss + kl_loss * alpha) / args.num_accum_steps                     # :85-86
            loss.backward()
            if (_step + 1) % args.num_accum_steps == 0:                                       # :89-93
                if record_first_grads is not None and not record_first_grads:
                    record_first_grads.update({n: sd_new[n].grad.detach().clone() for n in param_names})
                norm = torch.nn.utils.clip_grad_norm_(params, args.gradnorm_clip)
                tot_grad_norm += norm
                optim.step()
                optim.zero_grad()
            batch_losses.append(loss.detach().numpy())
            losses.append(loss.detach().numpy() * args.num_accum_steps)
            reward_losses.append(reward_loss.detach().numpy())
            kl_losses.append(kl_loss.detach().numpy())
        log.append(dict(epoch=epoch_num, mean_reward=float(np.mean(np.array(rewards))), grad_norm=float(tot_grad_norm),
                        loss=float(np.mean(losses)), reward_loss=float(np.mean(reward_losses)),
                        kl_loss=float(np.mean(kl_losses))))
    return batch_losses, log
