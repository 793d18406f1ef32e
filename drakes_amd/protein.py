"""The protein sampler's reverse loop on the wide-vocabulary HIP step:
`Interpolant.sample` of /root/reference/drakes_protein/fmif/fm_utils.py:86-139.

The denoiser there is ProteinMPNN, a message-passing network over backbone coordinates: it is the caller's torch
module and outside this path (as is classifier-free guidance's second forward, which is kept as the reference writes
it).  What runs on hand-written kernels is everything after the logits: temperature, mask handling, log-softmax,
posterior, Gumbel-argmax, carry-over, clean prediction — one launch per step (`drk_reverse_step_hard_wide`).
`sample_gradient` (:142-209) runs its hard steps on the same kernel and its Gumbel-softmax steps on
`drk_reverse_step_soft_wide_fwd/_bwd` under torch autograd (gradients reach the model through the logits and the soft
state it is fed).
"""
import torch
import torch.nn.functional as F

from . import autograd, ops

NUM_TOKENS = 22
MASK_TOKEN_INDEX = 21          # fmif/model_utils.py:10-13: 20 amino acids, 'X', then the mask token


class Interpolant:
    def __init__(self, cfg):
        self._cfg = cfg
        self.num_tokens = NUM_TOKENS
        self.neg_infinity = -1000000.0
        self._device = None

    def set_device(self, device):
        self._device = torch.device(device)

    @torch.no_grad()
    def sample(self, model, X, mask, chain_M, residue_idx, chain_encoding_all, cls=None, w=None, *, seed=None,
               sample_offset=0, uniforms=None):
        """Reference signature and return value: (pred_aatypes_1, prot_traj, clean_traj).  seed / sample_offset /
        uniforms(step) -> [B,L,22] are the noise controls of drakes_amd (the reference calls torch.rand_like, :14-18)."""
        num_batch, num_res = mask.shape
        dev = self._device if self._device is not None else mask.device
        if seed is None and uniforms is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        x = torch.full((num_batch, num_res), MASK_TOKEN_INDEX, dtype=torch.int64, device=dev)      # :96
        ts = torch.linspace(self._cfg.min_t, 1.0, self._cfg.num_timesteps)                         # :98-99
        t_1 = ts[0]
        prot_traj = [x.detach().cpu()]
        clean_traj = []
        pred = None
        for k, t_2 in enumerate(ts[1:]):
            d_t = t_2 - t_1
            if cls is not None:                                                                    # :107-112
                uncond = (2 * torch.ones(X.shape[0], device=X.device)).long()
                cond = (cls * torch.ones(X.shape[0], device=X.device)).long()
                out_u = model(X, x, mask, chain_M, residue_idx, chain_encoding_all, cls=uncond)
                out_c = model(X, x, mask, chain_M, residue_idx, chain_encoding_all, cls=cond)
                model_out = (1 + w) * out_c - w * out_u
            else:
                model_out = model(X, x, mask, chain_M, residue_idx, chain_encoding_all)
            logits = model_out.to(torch.float32).contiguous()
            u = uniforms(k) if uniforms is not None else None
            x, pred = ops.reverse_step_hard_wide(logits, x, float(d_t), float(1.0 - t_2), float(self._cfg.temp),
                                                 MASK_TOKEN_INDEX, seed or 0, k, sample_offset, u)
            clean_traj.append(pred.detach().cpu())
            prot_traj.append(x.cpu().detach())
            t_1 = t_2
        return pred, prot_traj, clean_traj

    def sample_gradient(self, model, X, mask, chain_M, residue_idx, chain_encoding_all, truncate_steps, gumbel_softmax_temp,
                        *, seed=None, sample_offset=0, uniforms=None):
        """Reference signature and return value (:142-209): (straight-through one-hot sample [B,L,22] with grad,
        last_x_list, move_chance_t_list, copy_flag_list)."""
        num_batch, num_res = mask.shape
        dev = self._device if self._device is not None else mask.device
        if seed is None and uniforms is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        V, M = self.num_tokens, MASK_TOKEN_INDEX
        x1 = F.one_hot(torch.full((num_batch, num_res), M, dtype=torch.int64, device=dev), num_classes=V).float()
        num_timesteps = self._cfg.num_timesteps
        ts = torch.linspace(self._cfg.min_t, 1.0, num_timesteps)
        t_1 = ts[0]
        last_x_list, move_chance_t_list, copy_flag_list = [], [], []
        for k, t_2 in enumerate(ts[1:]):
            d_t = t_2 - t_1
            u = uniforms(k) if uniforms is not None else None
            move_chance_t = 1.0 - t_1
            if k < num_timesteps - truncate_steps:                      # hard step, no graph kept (:171-178)
                with torch.no_grad():
                    logits = model(X, x1, mask, chain_M, residue_idx, chain_encoding_all).to(torch.float32).contiguous()
                    tok = x1.argmax(dim=-1)
                    nxt, _ = ops.reverse_step_hard_wide(logits, tok, float(d_t), float(1.0 - t_2), float(self._cfg.temp), M,
                                                        seed or 0, k, sample_offset, u, want_clean=False)
                    copy_flag = (tok != M).to(x1.dtype).unsqueeze(-1)
                    x2 = F.one_hot(nxt, num_classes=V).float()
                    x1 = x1.detach()
            else:                                                       # Gumbel-softmax step (:179-183)
                logits = model(X, x1, mask, chain_M, residue_idx, chain_encoding_all).to(torch.float32)
                x2, copy_flag = autograd.soft_step_wide_with_grad(logits, x1, float(d_t), float(1.0 - t_2),
                                                                  float(self._cfg.temp), float(gumbel_softmax_temp), M,
                                                                  seed or 0, k, sample_offset, u)
            last_x_list.append(x1)
            move_chance_t_list.append(move_chance_t + self._cfg.min_t)
            copy_flag_list.append(copy_flag)
            x1, t_1 = x2, t_2
        last_x_list.append(x1)
        move_chance_t_list.append(1.0 - t_1 + self._cfg.min_t)
        copy_flag_list.append(1 - x1[:, :, M].unsqueeze(-1))
        hard = F.one_hot(x1[:, :, :-1].argmax(dim=-1), num_classes=V).float()
        return x1 + (hard - x1).detach(), last_x_list, move_chance_t_list, copy_flag_list
