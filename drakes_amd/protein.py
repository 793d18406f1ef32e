"""The protein sampler's reverse loop on the wide-vocabulary HIP step:
`Interpolant.sample` of /root/reference/drakes_protein/fmif/fm_utils.py:86-139.

The denoiser there is ProteinMPNN, a message-passing network over backbone coordinates: it is the caller's torch
module and outside this path (as is classifier-free guidance's second forward, which is kept as the reference writes
it).  What runs on hand-written kernels is everything after the logits: temperature, mask handling, log-softmax,
posterior, Gumbel-argmax, carry-over, clean prediction — one launch per step (`drk_reverse_step_hard_wide`).
Only `sample` is here; `sample_gradient` (the Gumbel-softmax variant, :142-209) is not built yet.
"""
import torch

from . import ops

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
