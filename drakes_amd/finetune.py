"""Diffusion._sample_finetune_gradient / _ddpm_update_finetune_gradient on the HIP kernels
(reference: /root/reference/drakes_dna/diffusion_gosai_update.py:552-630).

The first num_steps - truncate_steps steps are hard steps without a graph (the reference builds
one and drops it, :616-619); the last truncate_steps steps are Gumbel-softmax steps whose chain
is differentiable w.r.t. the denoiser parameters (drakes_amd/autograd.py).
"""
import torch
import torch.nn.functional as F

from . import autograd, noise_schedule, ops


def _scalars(model, t, dt):
    """sigma_t [B], move_chance_t/s as the reference computes them (:557-568)."""
    sigma_t, _ = model.noise(t)
    sigma_s, _ = model.noise(t - dt)
    if sigma_t.ndim > 1:
        sigma_t = sigma_t.squeeze(-1)
    if sigma_s.ndim > 1:
        sigma_s = sigma_s.squeeze(-1)
    return sigma_t, 1 - torch.exp(-sigma_t), 1 - torch.exp(-sigma_s)


def ddpm_update_finetune_gradient(model, x, t, dt, copy_flag_temp, return_process=False, *, seed=None, step=0,
                                  sample_offset=0, uniforms=None):
    """One soft reverse step (:552-586).  x: int64 [B,L] or float32 [B,L,5]; t [B,1] (rows may differ)."""
    if x.ndim == 2 or x.shape[-1] != model.vocab_size:
        x = F.one_hot(x, num_classes=model.vocab_size).to(torch.float32)
    sigma_t, mct, mcs = _scalars(model, t, dt)
    mc_rows = None
    sig = model._process_sigma(sigma_t).to(x.device)
    if not (bool((mct == mct[0]).all()) and bool((mcs == mcs[0]).all())):
        mc_rows = torch.stack([mct, mcs], dim=-1).to(device=x.device, dtype=torch.float32).contiguous()
    else:
        sig = sig[:1]                                  # one shared value (all rows are equal)
    if seed is None and uniforms is None:
        seed = model._draw_seed()
    model.backbone.train_dtype = model._soft_dtype()
    logits = autograd.backbone_with_grad(model.backbone, x, sig, model._soft_dtype())
    x_next, flag = autograd.soft_step_with_grad(logits, x, float(mct[0]), float(mcs[0]),
                                                float(model.config.finetuning.gumbel_softmax_temp), copy_flag_temp,
                                                seed or 0, step, sample_offset, uniforms, mc_rows)
    if return_process:
        return x_next, x, sigma_t, mct[:, None, None], flag
    return x_next


def sample_finetune_gradient(model, num_steps=None, eps=1e-5, eval_sp_size=None, copy_flag_temp=None, *, seed=None,
                             sample_offset=0, uniforms=None):
    assert model.parameterization == 'subs' and model.sampler == 'ddpm'
    batch = model.config.loader.eval_batch_size if eval_sp_size is None else eval_sp_size
    if num_steps is None:
        num_steps = model.config.sampling.steps
    L = model.config.model.length
    dev = model.device
    if seed is None:
        seed = model._draw_seed()
    trunc = model.config.finetuning.truncate_steps
    temp = float(model.config.finetuning.gumbel_softmax_temp)
    table, _, _, _ = noise_schedule.step_table(model.noise, num_steps, eps)
    table = table.tolist()
    bb = model.backbone
    soft_bf16 = model._soft_dtype() == 'bf16'
    bb.train_dtype = model._soft_dtype()       # Diffusion.forward (the KL term of the fine-tune loop) follows it
    x = torch.full((batch, L), model.mask_index, dtype=torch.int64, device=dev)      # _sample_prior, on the device
    last_x_list, condt_list, move_chance_t_list, copy_flag_list = [], [], [], []
    # raw logits (with their graph) of every soft step, keyed by the identity of the state they were computed from: the
    # fine-tune loop's KL term evaluates the same model on the same tensor (finetune_loop.kl_of_step)
    model._soft_logits_cache = {}
    shared = None
    sig = None
    flat = None                                 # the parameters as one vector: one gradient per soft step, not 129
    sink = None                                 # bf16 soft steps: parameter gradients deferred to the end of the chain
    n_hard = max(num_steps - trunc, 0)
    first = 0
    if model.hip_dtype == 'bf16' and n_hard > 0 and L <= 224:
        # hip_dtype = 'bf16': the hard steps carry no gradient (:615-619), so they run as ONE launch of the fused bf16
        # sampler, which also records the state every step started from; the soft steps stay on the fp32 kernels.
        # Tokens may differ from the fp32 path's at bf16 level (as for _sample); parity tests use hip_dtype = 'fp32'.
        from . import fastpath
        with torch.no_grad():
            x, traj = fastpath.hard_steps_bf16(model, batch, table, n_hard, seed, sample_offset, uniforms)
            # the four record lists of the hard steps (:616-626), built for all steps at once and handed out as views:
            # a handful of launches instead of six per step
            tr = traj.to(torch.int64)                                                     # [n_hard, B, L]
            lx_all = F.one_hot(tr, num_classes=model.vocab_size).to(torch.float32)        # [n_hard, B, L, 5]
            cf_all = (tr != model.mask_index).to(torch.int64).unsqueeze(-1)               # [n_hard, B, L, 1]
            sched = torch.tensor([[row[1], row[2]] for row in table[:n_hard]], dtype=torch.float32,
                                 pin_memory=True).to(dev, non_blocking=True)   # (pinned: the copy does not wait for queued kernels)
            ct_all = sched[:, 0, None].expand(n_hard, batch).contiguous()
            mc_all = sched[:, 1, None, None, None].expand(n_hard, batch, 1, 1).contiguous()
            for i in range(n_hard):
                last_x_list.append(lx_all[i])
                condt_list.append(ct_all[i])
                condt_list[-1]._drk_uniform = float(table[i][1])
                move_chance_t_list.append(mc_all[i])
                copy_flag_list.append(cf_all[i])
        first = n_hard
    for i in range(first, num_steps):
        _, sigma_t, mc_t, mc_s = table[i]
        u = uniforms(i) if uniforms is not None else None
        condt = torch.full((batch,), sigma_t, dtype=torch.float32, device=dev)
        condt._drk_uniform = float(sigma_t)        # Diffusion.forward: every row holds this value
        mct = torch.full((batch, 1, 1), mc_t, dtype=torch.float32, device=dev)
        if i < num_steps - trunc:                                           # hard step, no graph (:615-619)
            with torch.no_grad():
                if model.time_conditioning or shared is None:
                    shared = model._shared_time_bias(sigma_t)
                logits = (bb.forward_x3 if model.hip_dtype == 'bf16x3' else bb.forward_f32)(x, shared, 0, model._work)
                x_next = ops.reverse_step_hard(logits, x, mc_t, mc_s, seed, i, sample_offset, u)
                last_x = F.one_hot(x, num_classes=model.vocab_size).to(torch.float32)
                copy_flag = (x != model.mask_index).to(x.dtype).unsqueeze(-1)
            x = x_next
        else:                                                               # soft step (:621, :552-586)
            if x.ndim == 2:
                x = F.one_hot(x, num_classes=model.vocab_size).to(torch.float32)
            if model.time_conditioning or sig is None:        # (one sigma tensor for all soft steps when it is constant)
                sig = torch.full((1,), sigma_t if model.time_conditioning else 0.0, dtype=torch.float32, device=dev)
            if flat is None and torch.is_grad_enabled() and getattr(model, 'hip_flat_grads', True):
                if soft_bf16:
                    sink, flat = autograd.grad_chain(bb)
                else:
                    flat = autograd.flat_params(bb)
            grad_on = torch.is_grad_enabled()
            logits = autograd.backbone_with_grad(bb, x, sig, bb.train_dtype, flat=flat if grad_on else None,
                                                 sink=sink if grad_on else None,
                                                 sigma_key=float(sigma_t) if model.time_conditioning else 0.0)
            last_x = x
            model._soft_logits_cache[id(x)] = (x, logits, bb._weights_key())
            x, copy_flag = autograd.soft_step_with_grad(logits, x, mc_t, mc_s, temp, copy_flag_temp, seed, i,
                                                        sample_offset, u)
        last_x_list.append(last_x)
        condt_list.append(condt)
        move_chance_t_list.append(mct)
        copy_flag_list.append(copy_flag)
    x_argmax = F.one_hot(x[:, :, :-1].argmax(dim=-1), num_classes=model.vocab_size - 1).to(torch.float32)
    return x[:, :, :-1] + (x_argmax - x[:, :, :-1]).detach(), last_x_list, condt_list, move_chance_t_list, copy_flag_list
