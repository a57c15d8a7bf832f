"""Noise schedule of the masked diffusion (reference noise_schedule.py:126-152, only the live
`loglinear` type) and the per-step move-chance table the SVDD kernels consume."""
import torch
from torch import nn


class LogLinearNoise(nn.Module):
    """total_noise(t) = -log1p(-(1-eps) t), so that 1 - exp(-sigma(t)) = (1-eps) t
    (reference noise_schedule.py:126-145). forward(t) -> (total_noise, rate_noise) like
    `Noise.forward` (:41-43)."""

    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps

    def rate_noise(self, t):
        return (1 - self.eps) / (1 - (1 - self.eps) * t)

    def total_noise(self, t):
        return -torch.log1p(-(1 - self.eps) * t)

    def forward(self, t):
        return self.total_noise(t), self.rate_noise(t)


def get_noise(config):
    if config.noise.type != "loglinear":
        raise ValueError(f"{config.noise.type}: only the loglinear schedule is live in the reference "
                         "(configs_gosai/config_gosai.yaml:8)")
    return LogLinearNoise()


def move_chance_table(noise, num_steps, eps=1e-5):
    """The scalars every `_ddpm_update_*` recomputes (diffusion_gosai.py:1036-1038, 1176-1187),
    evaluated once on the host with the same fp32 torch ops, in the same order:
      timesteps = linspace(1, eps, S+1); dt = (1-eps)/S
      sigma_t = noise(t); sigma_s = noise(t - dt); move_chance = 1 - exp(-sigma)
    Returns (table fp32 [S,3] = (mct, mcs, mct - mcs), timesteps fp32 [S+1], dt float)."""
    timesteps = torch.linspace(1, eps, num_steps + 1)
    dt = (1 - eps) / num_steps
    t = timesteps[:num_steps].view(-1, 1)
    sigma_t, _ = noise(t)
    sigma_s, _ = noise(t - dt)
    mct = 1 - torch.exp(-sigma_t.squeeze(-1))
    mcs = 1 - torch.exp(-sigma_s.squeeze(-1))
    return torch.stack([mct, mcs, mct - mcs], dim=1), timesteps, dt
