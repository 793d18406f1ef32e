"""Noise schedule of the reverse sampler.

Mirrors the interface of /root/reference/drakes_dna/noise_schedule.py:37-58,126-151 for the
one schedule the sampler configuration selects (configs_gosai/noise/loglinear.yaml).  The
schedule is evaluated on the HOST: every per-step quantity of `_sample` depends only on the
step index (diffusion_gosai_update.py:521-528), so the sampler turns it into a table of fp32
scalars once per call and hands the values to the kernels as arguments.
"""
import torch
import torch.nn as nn


class LogLinearNoise(nn.Module):
    """total_noise(t) = -log1p(-(1-eps) t); 1 - exp(-total_noise) runs from 0 to ~1."""

    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps
        self.sigma_max = self.total_noise(torch.tensor(1.0))
        self.sigma_min = self.eps + self.total_noise(torch.tensor(0.0))

    def rate_noise(self, t):
        return (1 - self.eps) / (1 - (1 - self.eps) * t)

    def total_noise(self, t):
        return -torch.log1p(-(1 - self.eps) * t)

    def forward(self, t):
        return self.total_noise(t), self.rate_noise(t)


def get_noise(config, dtype=torch.float32):
    if config.noise.type == 'loglinear':
        return LogLinearNoise()
    raise ValueError(f'{config.noise.type} is not a noise schedule of the sampling hot path '
                     '(only loglinear is in scope; see DESIGN.md)')


def step_table(noise, num_steps, eps=1e-5):
    """Host fp32 table, one row per reverse step: (t, sigma_t, move_chance_t, move_chance_s),
    plus sigma at the final time for the noise-removal forward.

    Same torch ops, same order as _sample / _ddpm_update (diffusion_gosai_update.py:463-474,
    :521-528, :541-547) so the values are bit-identical to what the reference feeds its step.
    """
    timesteps = torch.linspace(1, eps, num_steps + 1)
    dt = (1 - eps) / num_steps
    t = timesteps[:num_steps].reshape(-1, 1)
    sigma_t = noise(t)[0].squeeze(-1)
    sigma_s = noise(t - dt)[0].squeeze(-1)
    mc_t = 1 - torch.exp(-sigma_t)
    mc_s = 1 - torch.exp(-sigma_s)
    sigma_last = noise(timesteps[-1].reshape(1, 1))[0].reshape(())
    return torch.stack([t.squeeze(-1), sigma_t, mc_t, mc_s], dim=1), sigma_last, timesteps, dt
