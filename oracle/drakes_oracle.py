"""TEST INFRASTRUCTURE — the CPU oracle for the DRAKES reverse-sampling hot path.

A functional, CPU-only (torch fp32) restatement of the reference algorithm.  It is
the CHECKER for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Nothing under drakes_amd/ imports it, and the
product path raises when the HIP library is missing instead of coming here.

Pinned: oracle/make_golden.py imports the real reference class (with stand-ins for
the third-party packages this image lacks) and checks every function below against
it with the same weights and the same injected uniforms, then writes the fixtures
in tests/golden/.  tests/test_oracle_golden.py re-checks the oracle against those
fixtures on every run, so the oracle cannot drift from the reference unnoticed.

The reference uses stock torch ops; using the same ops in the same order makes this
restatement bit-identical to the reference on the same host (make_golden.py asserts
exact equality), which is the property the token-exactness tests lean on.

file:line citations are into /root/reference/drakes_dna/.
"""
import numpy as np
import torch
import torch.nn.functional as F

MASK_INDEX = 4        # diffusion_gosai_update.py:82-88 (vocab 4 -> 5, mask is the last id)
VOCAB = 5
NEG_INFINITY = -1000000.0   # diffusion_gosai_update.py:127
NOISE_EPS = 1e-3      # noise_schedule.py:134


# --------------------------------------------------------------------------- schedule
def total_noise(t):
    """LogLinearNoise.total_noise — noise_schedule.py:143-144."""
    return -torch.log1p(-(1 - NOISE_EPS) * t)


def timestep_grid(num_steps, eps=1e-5):
    """`timesteps`, `dt` of _sample — diffusion_gosai_update.py:521-523."""
    return torch.linspace(1, eps, num_steps + 1), (1 - eps) / num_steps


def step_scalars(num_steps, eps=1e-5):
    """fp32 table [num_steps, 4] = (t, sigma_t, move_chance_t, move_chance_s).

    Follows _ddpm_update — diffusion_gosai_update.py:463-474 — for one batch row
    (all rows are equal: :527-528)."""
    ts, dt = timestep_grid(num_steps, eps)
    t = ts[:num_steps].reshape(-1, 1) * torch.ones(1, 1)
    sigma_t = total_noise(t).squeeze(-1)
    sigma_s = total_noise(t - dt).squeeze(-1)
    mc_t = 1 - torch.exp(-sigma_t)
    mc_s = 1 - torch.exp(-sigma_s)
    return torch.stack([t.squeeze(-1), sigma_t, mc_t, mc_s], dim=1)


# --------------------------------------------------------------------------- denoiser
def n_layers_of(sd):
    n = 0
    while f'backbone.convs.{n}.weight' in sd:
        n += 1
    return n


def layer_dilation(i, n_layers):
    """convs = 5 layer types, each repeated num_cnn_stacks times consecutively —
    models/dnaconv.py:53-58."""
    return (1, 1, 4, 16, 64)[i // (n_layers // 5)]


def time_embedding(sd, sigma):
    """relu(Linear(GaussianFourierProjection(sigma))) — models/dnaconv.py:19-21,50,82."""
    W = sd['backbone.time_embedder.0.W']
    proj = sigma[:, None] * W[None, :] * 2 * np.pi
    four = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)
    return F.relu(F.linear(four, sd['backbone.time_embedder.1.weight'],
                           sd['backbone.time_embedder.1.bias']))


def cnn_forward(sd, seq, sigma, return_feats=False, preacts=None):
    """CNNModel.forward — models/dnaconv.py:74-110 (clean_data=False, no cfg, p_drop=0).

    seq: int64 [B,L] or float [B,L,5]; sigma: [B].  Returns logits [B,L,5]."""
    if not (seq.ndim > 2 and seq.shape[-1] == VOCAB):
        seq = F.one_hot(seq, num_classes=VOCAB).float()
    temb = time_embedding(sd, sigma)
    feat = F.relu(F.conv1d(seq.permute(0, 2, 1), sd['backbone.linear.weight'],
                           sd['backbone.linear.bias'], padding=4))
    n = n_layers_of(sd)
    feats = [feat]
    for i in range(n):
        d = layer_dilation(i, n)
        tb = F.linear(temb, sd[f'backbone.time_layers.{i}.dense.weight'],
                      sd[f'backbone.time_layers.{i}.dense.bias'])
        h = feat + tb[:, :, None]
        C = h.shape[1]
        h = F.layer_norm(h.permute(0, 2, 1), (C,), sd[f'backbone.norms.{i}.weight'],
                         sd[f'backbone.norms.{i}.bias'])
        c = F.conv1d(h.permute(0, 2, 1), sd[f'backbone.convs.{i}.weight'],
                     sd[f'backbone.convs.{i}.bias'], dilation=d, padding=4 * d)
        if preacts is not None:      # tests: how close any ReLU argument sits to its kink
            preacts.append(c.detach())
        h = F.relu(c)
        feat = h + feat
        feats.append(feat)
    c = F.conv1d(feat, sd['backbone.final_conv.0.weight'], sd['backbone.final_conv.0.bias'])
    if preacts is not None:
        preacts.append(c.detach())
    y = F.conv1d(F.relu(c), sd['backbone.final_conv.2.weight'], sd['backbone.final_conv.2.bias'])
    logits = y.permute(0, 2, 1)
    if return_feats:
        return logits, feats
    return logits


# --------------------------------------------------------------------------- SUBS
def subs_parameterization(logits, xt):
    """_subs_parameterization — diffusion_gosai_update.py:243-253, without the in-place
    writes: same values, and (as there) no gradient flows through unmasked rows."""
    shift = torch.zeros(VOCAB, dtype=logits.dtype)
    shift[MASK_INDEX] = NEG_INFINITY
    logits = logits + shift
    logp = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
    if xt.ndim > 2 and xt.shape[-1] == VOCAB:
        xt = xt.argmax(dim=-1)
    unmasked = (xt != MASK_INDEX)
    held = torch.full_like(logp, NEG_INFINITY).scatter(-1, xt.unsqueeze(-1), 0.0)
    return torch.where(unmasked.unsqueeze(-1), held, logp)


def model_forward(sd, x, sigma, time_conditioning=False):
    """Diffusion.forward + _process_sigma — diffusion_gosai_update.py:255-275."""
    if sigma.ndim > 1:
        sigma = sigma.squeeze(-1)
    if not time_conditioning:
        sigma = torch.zeros_like(sigma)
    return subs_parameterization(cnn_forward(sd, x, sigma), x)


# --------------------------------------------------------------------------- step
def gumbel_norm(u):
    """1e-10 - log(U + 1e-10) — diffusion_gosai_update.py:25-27."""
    return 1e-10 - (u + 1e-10).log()


def sample_categorical(q, u):
    """_sample_categorical with the uniform draw injected — :24-28 (divide, then argmax)."""
    return (q / gumbel_norm(u)).argmax(dim=-1)


def sample_categorical_gradient(q, u, temp=1.0):
    """_sample_categorical_gradient — :30-35."""
    return F.softmax((torch.log(q) - torch.log(gumbel_norm(u))) / temp, 2)


def posterior(log_p_x0, mc_t, mc_s):
    """q_xs of _ddpm_update — :478-480.  mc_* broadcast as [B,1,1] or scalars."""
    q = log_p_x0.exp() * (mc_t - mc_s)
    mask_col = torch.zeros(VOCAB, dtype=torch.bool)
    mask_col[MASK_INDEX] = True
    return torch.where(mask_col, torch.as_tensor(mc_s, dtype=q.dtype).expand_as(q[..., :1]), q)


def hard_step_from_logp(log_p_x0, x, u, mc_t, mc_s):
    """Tail of _ddpm_update — :478-486."""
    xs = sample_categorical(posterior(log_p_x0, mc_t, mc_s), u)
    copy_flag = (x != MASK_INDEX).to(x.dtype)
    return copy_flag * x + (1 - copy_flag) * xs


def hard_step_from_logits(logits, x, u, mc_t, mc_s):
    """SUBS + posterior + Gumbel-argmax + carry-over for raw backbone logits."""
    return hard_step_from_logp(subs_parameterization(logits, x), x, u, mc_t, mc_s)


def soft_step_from_logp(log_p_x0, x_soft, u, mc_t, mc_s, temp=1.0, copy_flag_temp=None):
    """Tail of _ddpm_update_finetune_gradient — :572-586.  Returns (x_next, soft_copy_flag)."""
    xs = sample_categorical_gradient(posterior(log_p_x0, mc_t, mc_s), u, temp)
    if copy_flag_temp is not None:
        flag = torch.sigmoid((1 - x_soft[:, :, MASK_INDEX].unsqueeze(-1)) / copy_flag_temp)
    else:
        flag = 1 - x_soft[:, :, MASK_INDEX].unsqueeze(-1)
    return flag * x_soft + (1 - flag) * xs, flag


class Schedule:
    """Per-step tensors exactly as _sample/_ddpm_update build them (all rows equal)."""

    def __init__(self, num_steps, eps=1e-5):
        self.num_steps = num_steps
        self.timesteps, self.dt = timestep_grid(num_steps, eps)

    def at(self, i, batch):
        t = self.timesteps[i] * torch.ones(batch, 1)
        sigma_t = total_noise(t).squeeze(-1)
        sigma_s = total_noise(t - self.dt).squeeze(-1)
        mc_t = (1 - torch.exp(-sigma_t))[:, None, None]
        mc_s = (1 - torch.exp(-sigma_s))[:, None, None]
        return sigma_t, mc_t, mc_s


# --------------------------------------------------------------------------- samplers
def sample(sd, batch, length, num_steps, uniform_fn, eps=1e-5, time_conditioning=False,
           noise_removal=True, trace=None):
    """Diffusion._sample, ddpm predictor — diffusion_gosai_update.py:506-550.

    uniform_fn(step, (B,L,5)) -> float32 tensor replaces torch.rand_like (:27)."""
    with torch.no_grad():
        x = MASK_INDEX * torch.ones(batch, length, dtype=torch.int64)          # :438-440
        sch = Schedule(num_steps, eps)
        for i in range(num_steps):
            sigma_t, mc_t, mc_s = sch.at(i, batch)
            logp = model_forward(sd, x, sigma_t, time_conditioning)
            x = hard_step_from_logp(logp, x, uniform_fn(i, tuple(logp.shape)), mc_t, mc_s)
            if trace is not None:
                trace.append(x.clone())
        if noise_removal:                                                      # :541-549
            t = sch.timesteps[-1] * torch.ones(batch, 1)
            logp = model_forward(sd, x, total_noise(t), time_conditioning)
            x = logp[:, :, :-1].argmax(dim=-1)
        return x


def sample_finetune_gradient(sd, batch, length, num_steps, truncate_steps, uniform_fn,
                             eps=1e-5, temp=1.0, copy_flag_temp=None, time_conditioning=False):
    """Diffusion._sample_finetune_gradient — diffusion_gosai_update.py:589-630.

    Gradients flow to whichever tensors of `sd` have requires_grad=True."""
    x = MASK_INDEX * torch.ones(batch, length, dtype=torch.int64)
    sch = Schedule(num_steps, eps)
    last_x_list, condt_list, mct_list, flag_list = [], [], [], []
    for i in range(num_steps):
        sigma_t, mc_t, mc_s = sch.at(i, batch)
        if i < num_steps - truncate_steps:                                     # :615-619
            logp = model_forward(sd, x, sigma_t, time_conditioning)
            u = uniform_fn(i, tuple(logp.shape))
            last_x = F.one_hot(x, num_classes=VOCAB).to(torch.float32).detach()
            copy_flag = (x != MASK_INDEX).to(x.dtype).unsqueeze(-1)
            x = hard_step_from_logp(logp, x, u, mc_t, mc_s).detach()
        else:                                                                  # :621, :552-586
            if x.ndim == 2:
                x = F.one_hot(x, num_classes=VOCAB).to(torch.float32)
            logp = model_forward(sd, x, sigma_t, time_conditioning)
            u = uniform_fn(i, tuple(logp.shape))
            last_x = x
            x, copy_flag = soft_step_from_logp(logp, x, u, mc_t, mc_s, temp, copy_flag_temp)
        last_x_list.append(last_x)
        condt_list.append(sigma_t)
        mct_list.append(mc_t)
        flag_list.append(copy_flag)
    if x.ndim == 2:  # truncate_steps == 0: the reference would fail at :628; keep its precondition
        raise ValueError('truncate_steps must be >= 1')
    hard = F.one_hot(x[:, :, :-1].argmax(dim=-1), num_classes=VOCAB - 1).to(torch.float32)
    out = x[:, :, :-1] + (hard - x[:, :, :-1]).detach()                        # :628-630
    return out, last_x_list, condt_list, mct_list, flag_list


def kl_step(logp_new, logp_old, last_x, copy_flag, mc_t_scalar):
    """Per-step KL term of fine_tune — finetune_reward_bp.py:67-74.  Inputs are full
    5-wide log-probs; returns [B]."""
    lp, lo = logp_new[:, :, :-1], logp_old[:, :, :-1]
    p, po = lp.exp(), lo.exp()
    kl = copy_flag * (-p + po + p * (lp - lo)) / mc_t_scalar
    return (kl * last_x[:, :, :-1]).sum((1, 2))


# --------------------------------------------------------------------------- likelihood
def q_xt(x, move_chance, u):
    """Diffusion.q_xt with the uniform draw injected — diffusion_gosai_update.py:425-436.  u: [B,L] in [0,1)."""
    return torch.where(u < move_chance, MASK_INDEX, x)


def get_likelihood(sd, x0, num_steps, uniform_fn, eps=1e-5, n_samples=1, time_conditioning=False):
    """Diffusion.get_likelihood — diffusion_gosai_update.py:885-926.  uniform_fn(sample, step) -> float32 [B,L]."""
    with torch.no_grad():
        timesteps, dt = timestep_grid(num_steps, eps)
        per_sample = []
        for s in range(n_samples):
            at_time = []
            for i in range(num_steps):
                t = timesteps[i] * torch.ones(x0.shape[0], 1)
                sigma_t = total_noise(t).squeeze(-1)
                sigma_s = total_noise(t - dt).squeeze(-1)
                mc_t = (1 - torch.exp(-sigma_t))[:, None]
                mc_s = (1 - torch.exp(-sigma_s))[:, None]
                multiplier = (mc_t - mc_s) / mc_t
                xt = q_xt(x0, mc_t, uniform_fn(s, i))
                out = model_forward(sd, xt, sigma_t, time_conditioning)
                at_time.append(out.gather(-1, x0[..., None]).squeeze(-1) * multiplier)
            per_sample.append(torch.stack(at_time, dim=0).sum(dim=0).sum(dim=-1))
        return torch.stack(per_sample, dim=0).mean(dim=0)


# --------------------------------------------------------------------------- wide vocabulary (protein, V = 22)
def hard_step_wide(logits, x, u, d_t, move_chance_s, temp=1.0, mask_index=21, neg_infinity=-1000000.0):
    """One reverse step of Interpolant.sample — /root/reference/drakes_protein/fmif/fm_utils.py:113-134 — from the
    model's logits [B,L,V]: returns (x_next int64 [B,L], pred_clean int64 [B,L]).  u: injected uniforms [B,L,V]."""
    pred_wo_mask = logits.clone()
    pred_wo_mask[:, :, mask_index] = -1e9
    pred_clean = torch.argmax(pred_wo_mask, dim=-1)
    lg = logits.clone()
    lg[:, :, mask_index] = neg_infinity
    lg = lg / temp - torch.logsumexp(lg / temp, dim=-1, keepdim=True)
    unmasked = (x != mask_index)
    lg[unmasked] = neg_infinity
    lg[unmasked, x[unmasked]] = 0
    q = lg.exp() * d_t
    q[:, :, mask_index] = move_chance_s
    xs = (q / gumbel_norm(u)).argmax(dim=-1)
    copy_flag = unmasked.to(x.dtype)
    return (x * copy_flag + xs * (1 - copy_flag)).long(), pred_clean


def sample_gradient_wide(model_fn, batch, length, num_timesteps, min_t, temp, truncate_steps, gumbel_temp, uniform_fn,
                         num_tokens=22, mask_index=21, neg_infinity=-1000000.0):
    """Interpolant.sample_gradient — /root/reference/drakes_protein/fmif/fm_utils.py:142-209.  model_fn(k, x_soft) ->
    logits [B,L,V] (stands for ProteinMPNN); uniform_fn(k) -> [B,L,V].  Returns the reference's 4-tuple."""
    x1 = F.one_hot(torch.full((batch, length), mask_index), num_classes=num_tokens).float()
    ts = torch.linspace(min_t, 1.0, num_timesteps)
    t_1 = ts[0]
    last_x_list, mct_list, flag_list = [], [], []
    for k, t_2 in enumerate(ts[1:]):
        d_t = t_2 - t_1
        lg = model_fn(k, x1).clone()
        lg[:, :, mask_index] = neg_infinity
        lg = lg / temp - torch.logsumexp(lg / temp, dim=-1, keepdim=True)
        am = x1.argmax(dim=-1)
        unmasked = (am != mask_index)
        held = torch.full_like(lg, neg_infinity).scatter(-1, am.unsqueeze(-1), 0.0)
        lg = torch.where(unmasked.unsqueeze(-1), held, lg)
        move_chance_t = 1.0 - t_1
        move_chance_s = 1.0 - t_2
        q = lg.exp() * d_t
        col = torch.zeros(num_tokens, dtype=torch.bool)
        col[mask_index] = True
        q = torch.where(col, move_chance_s.expand_as(q[..., :1]), q)
        u = uniform_fn(k)
        if k < num_timesteps - truncate_steps:
            xs = F.one_hot((q / gumbel_norm(u)).argmax(dim=-1), num_classes=num_tokens).float()
            flag = unmasked.to(x1.dtype).unsqueeze(-1)
            x2 = (x1 * flag + xs * (1 - flag)).detach()
            x1 = x1.detach()
        else:
            q = q + 1e-8
            xs = F.softmax((torch.log(q) - torch.log(gumbel_norm(u))) / gumbel_temp, 2)
            flag = 1 - x1[:, :, mask_index].unsqueeze(-1)
            x2 = x1 * flag + xs * (1 - flag)
        last_x_list.append(x1)
        mct_list.append(move_chance_t + min_t)
        flag_list.append(flag)
        x1, t_1 = x2, t_2
    last_x_list.append(x1)
    mct_list.append(1.0 - t_1 + min_t)
    flag_list.append(1 - x1[:, :, mask_index].unsqueeze(-1))
    hard = F.one_hot(x1[:, :, :-1].argmax(dim=-1), num_classes=num_tokens).float()
    return x1 + (hard - x1).detach(), last_x_list, mct_list, flag_list
