"""The DRAKES fine-tune loop on the HIP path: `fine_tune` of the reference
(/root/reference/drakes_dna/finetune_reward_bp.py:17-123) — reward back-propagated through the last K Gumbel-softmax
reverse steps plus a per-step KL term against the frozen pre-trained model, gradient accumulation, clipping, Adam.

Same arguments and return value as the reference function.  Differences, all additive:
  * the reference reads the module globals `log_path` / `save_path` and logs to wandb; here they are keyword
    arguments (None = no file output) and there is no wandb;
  * `seed_of_call(i)` fixes the Philox seed of the i-th `_sample_finetune_gradient` call (default: drawn from torch's
    generator, so torch.manual_seed reproduces a run);
  * `old_model`'s parameters are frozen (requires_grad False) for the duration of the call: the reference accumulates
    gradients on them that nothing reads; the gradient THROUGH the old model into last_x is kept (:68);
  * `on_update(model, epoch)` is called before every optimizer step with the reduced, unclipped gradients in
    place (tests, logging);
  * `skip_hard_step_kl` (default on): the KL term of a HARD step is identically 0.0 with zero gradient, in floating
    point too — there last_x is one-hot, so masked positions are multiplied by last_x[..., :4] = 0 and unmasked ones
    have log_p = log_p_old = the SUBS constants (0 at the held token, -1e6 elsewhere), hence p = p_old and
    -p + p_old + p * (log_p - log_p_old) = 0 exactly.  Those steps (78 of 128 by default) append an exact zero instead of
    running two forwards and a backward each; losses and gradients are bit-identical either way (tested on F6 a);
  * `reuse_sampling_forward` (default on): see kl_of_step — the KL term of a soft step reuses the logits the sampler
    computed from the same tensor instead of a second forward/backward of the new model;
  * `kl_streams` (default 2; 0 = off): the frozen model's forwards of the KL term run round-robin on that many side
    streams, and autograd runs their input-gradient launches there too, beside the new model's backward chain (at
    fine-tune batch sizes one launch fills at most half of the chip); same values;
  * under torch.distributed (drakes_amd.dp) every rank runs its own shard (`args.batch_size` rows per rank); the flat
    gradient bucket is all-reduced once, after the last accumulation micro-step and BEFORE clipping (:89-92).
The reward models are the caller's torch modules (the gReLU model in the reference, oracle.py:230-262): outside the
path, called as they are.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import autograd, dp


def kl_of_step(new_model, old_model, last_x, condt, move_chance_t, copy_flag, reuse_sampling_forward=True, old_logp=None):
    """kl_div [B] of one recorded reverse step — finetune_reward_bp.py:63-74, one fused kernel each way.

    On soft steps last_x carries the sampling graph, so the reference's gradient also runs back through
    old_model.forward(last_x) (:68) into the chain; the frozen model then passes only its input gradient on.  On hard
    steps last_x is a constant and the frozen model runs without a graph.

    reuse_sampling_forward: `new_model.forward(last_x, condt)` (:67) repeats, on the very same tensor and weights, the
    denoiser forward that `_sample_finetune_gradient` ran to take that step.  Its raw logits (and their graph) are
    still there (finetune.sample_finetune_gradient keeps them), so the KL term applies SUBS to them instead of running
    the network again: one forward and one backward less per soft step; the gradient reaches the weights through the
    shared node (same value up to fp32 summation order)."""
    hit = getattr(new_model, '_soft_logits_cache', {}).get(id(last_x)) if reuse_sampling_forward else None
    if hit is not None and hit[0] is last_x and hit[2] == new_model.backbone._weights_key():
        log_p_x0 = new_model._subs_parameterization(logits=hit[1], xt=last_x)
    else:
        log_p_x0 = new_model.forward(last_x, condt)
    if old_logp is not None:
        log_p_x0_old = old_logp
    elif last_x.requires_grad:
        log_p_x0_old = old_model.forward(last_x, condt)
    else:
        with torch.no_grad():
            log_p_x0_old = old_model.forward(last_x, condt)
    return autograd.kl_step_with_grad(log_p_x0, log_p_x0_old, last_x, copy_flag, float(move_chance_t.reshape(-1)[0]))


def old_model_forwards(old_model, steps, last_x_list, condt_list, n_streams):
    """{step: old_model.forward(last_x[step], condt[step])} for the given steps, run round-robin on `n_streams` side
    streams: the forwards are independent of each other, and at fine-tune batch sizes (one workgroup per sequence, 32-128
    sequences) each fills at most half of the chip.  autograd runs each one's backward on the stream of its forward, so
    the frozen model's input-gradient launches overlap the new model's backward chain the same way.  The caller's
    stream waits for the side streams before this returns."""
    cur = torch.cuda.current_stream()
    streams = _side_streams(cur.device, n_streams)
    out = {}
    # whatever the frozen model caches per weight version (kernel-layout copies of its weights) is built HERE, on the
    # caller's stream: built inside the first side-stream forward, the second stream would read it unordered
    bb = old_model.backbone
    pk = bb.packed()
    soft = old_model.hip_soft_dtype or old_model.hip_dtype
    if soft == 'bf16' and getattr(bb, 'width', 128) == 128:
        from . import fastpath
        fastpath._packed(bb)
        fastpath._packed_T(bb)
    else:
        for k in ('conv_w', 'conv_wT', 'head0_w', 'head0_wT'):
            pk[k]
        if soft == 'bf16x3':
            bb.packed_x3()
    for s in streams:
        s.wait_stream(cur)
    for k, t in enumerate(steps):
        with torch.cuda.stream(streams[k % n_streams]):
            if last_x_list[t].requires_grad:
                out[t] = old_model.forward(last_x_list[t], condt_list[t])
            else:
                with torch.no_grad():
                    out[t] = old_model.forward(last_x_list[t], condt_list[t])
            out[t].record_stream(cur)
    for s in streams:
        cur.wait_stream(s)
    return out


_STREAMS = {}


def _side_streams(device, n):
    key = (device.index, n)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _STREAMS[key]


def fine_tune(new_model, new_model_y, new_model_y_eval, old_model, args, eps=1e-5, *, log_path=None, save_path=None,
              seed_of_call=None, verbose=True, on_update=None, skip_hard_step_kl=True,
              reuse_sampling_forward=True, kl_streams=2):
    if log_path is not None:
        with open(log_path, 'w') as f:
            f.write(args.__repr__() + '\n')

    new_model.config.finetuning.truncate_steps = args.truncate_steps                          # :22-23
    new_model.config.finetuning.gumbel_softmax_temp = args.gumbel_temp
    new_model.train()
    torch.set_grad_enabled(True)
    old_flags = [p.requires_grad for p in old_model.parameters()]
    for p in old_model.parameters():
        p.requires_grad_(False)
    optim = torch.optim.Adam(new_model.parameters(), lr=args.learning_rate)                   # :27
    bucket = dp.GradBucket(new_model)           # every .grad is a view of one flat fp32 buffer (one all-reduce, one norm)
    bucket.zero_()
    batch_losses = []
    batch_rewards = []
    call = 0
    rank, world = dp.rank(), dp.world()

    for epoch_num in range(args.num_epochs):                                                  # :31
        rewards, rewards_eval, losses, reward_losses, kl_losses = [], [], [], [], []
        tot_grad_norm = 0.0
        new_model.train()
        for _step in range(args.num_accum_steps):                                             # :39
            kw = {}
            if seed_of_call is not None:
                kw['seed'] = seed_of_call(call)
            call += 1
            sample, last_x_list, condt_list, move_chance_t_list, copy_flag_list = new_model._sample_finetune_gradient(
                eval_sp_size=args.batch_size, copy_flag_temp=args.copy_flag_temp,
                sample_offset=rank * args.batch_size, **kw)                                   # :40, [bsz, seqlen, 4]
            sample2 = torch.transpose(sample, 1, 2)
            preds = new_model_y(sample2).squeeze(-1)                                          # [bsz, 3]
            reward = preds[:, 0]

            with torch.no_grad():                                                             # logging only (:45-55)
                sample_argmax = 1.0 * F.one_hot(torch.argmax(sample, 2), num_classes=4)
                sample_argmax = torch.transpose(sample_argmax, 1, 2)
                rewards.append(new_model_y(sample_argmax).squeeze(-1)[:, 0].cpu().numpy())
                rewards_eval.append(new_model_y_eval(sample_argmax).squeeze(-1)[:, 0].cpu().numpy())

            total_kl = []
            kl_steps = [t for t in range(args.total_num_steps)                                # :60-75
                        if not (args.truncate_kl and t < args.total_num_steps - args.truncate_steps)]
            run_steps = [t for t in kl_steps if not (skip_hard_step_kl and copy_flag_list[t].dtype == torch.int64)]
            old_logps = {}
            if kl_streams and sample.is_cuda and len(run_steps) > 1:
                old_logps = old_model_forwards(old_model, run_steps, last_x_list, condt_list, kl_streams)
            for random_t in kl_steps:
                if skip_hard_step_kl and copy_flag_list[random_t].dtype == torch.int64:
                    total_kl.append(torch.zeros(args.batch_size, dtype=torch.float32, device=sample.device))
                    continue
                total_kl.append(kl_of_step(new_model, old_model, last_x_list[random_t], condt_list[random_t],
                                           move_chance_t_list[random_t], copy_flag_list[random_t], reuse_sampling_forward,
                                           old_logps.get(random_t)))

            if epoch_num < args.alpha_schedule_warmup:                                        # :77-81, linear warm-up
                current_alpha = (epoch_num + 1) / args.alpha_schedule_warmup * args.alpha
            else:
                current_alpha = args.alpha

            kl_loss = torch.stack(total_kl, 1).sum(1).mean()                                  # :83-86
            reward_loss = -torch.mean(reward)
            loss = reward_loss + kl_loss * current_alpha
            loss = loss / args.num_accum_steps

            loss.backward()
            if (_step + 1) % args.num_accum_steps == 0:                                       # :89-93
                bucket.allreduce_mean_()
                if on_update is not None:
                    on_update(new_model, epoch_num)
                norm = bucket.clip_grad_norm_(args.gradnorm_clip)
                tot_grad_norm += float(norm)
                optim.step()
                bucket.zero_()                  # optim.zero_grad() would drop the views

            batch_losses.append(loss.cpu().detach().numpy())
            batch_rewards.append(torch.mean(reward).cpu().detach().numpy())
            losses.append(loss.cpu().detach().numpy() * args.num_accum_steps)
            reward_losses.append(reward_loss.cpu().detach().numpy())
            kl_losses.append(kl_loss.cpu().detach().numpy())

        rewards, rewards_eval = np.array(rewards), np.array(rewards_eval)
        line = (f'Epoch {epoch_num} Mean reward {np.mean(rewards)} Mean reward eval {np.mean(rewards_eval)} '
                f'Mean grad norm {tot_grad_norm} Mean loss {np.mean(losses)} Mean reward loss {np.mean(reward_losses)} '
                f'Mean kl loss {np.mean(kl_losses)}')
        if verbose and rank == 0:
            print(line, flush=True)
        if log_path is not None and rank == 0:
            with open(log_path, 'a') as f:
                f.write(line + '\n')
        if save_path is not None and rank == 0 and (epoch_num + 1) % args.save_every_n_epochs == 0:   # :115-118
            torch.save(new_model.state_dict(), os.path.join(save_path, f'model_{epoch_num}.ckpt'))
    for p, f in zip(old_model.parameters(), old_flags):
        p.requires_grad_(f)
    fine_tune.last_log = dict(tot_grad_norm=tot_grad_norm, mean_reward=float(np.mean(rewards)),
                              mean_kl=float(np.mean(kl_losses)))
    return batch_losses
