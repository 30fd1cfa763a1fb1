"""Noise schedules and the per-step scalars of the ancestral sampler, computed on the HOST with torch in
the reference's own operation order (torch.linspace / log1p / exp must not be re-derived
in device code — GPU libm differs in ulps).

  LogLinearNoise  /root/reference/slm/utils/noise_utils.py:188-213
  CosineNoise     /root/reference/slm/utils/noise_utils.py:122-135   (fallback, model.py:345-347)
  timesteps/dt    /root/reference/slm/models/model.py:564-567
  sigma, move chance  model.py:584-595
  sinusoid of sigma   /root/reference/slm/models/net.py:497-517
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


class Noise:
    def total_noise(self, t):
        raise NotImplementedError

    def rate_noise(self, t):
        raise NotImplementedError

    def __call__(self, t):
        return self.total_noise(t), self.rate_noise(t)


class LogLinearNoise(Noise):
    def __init__(self, eps: float = 1e-3):
        self.eps = eps

    def total_noise(self, t):
        return -torch.log1p(-(1 - self.eps) * t)

    def rate_noise(self, t):
        return (1 - self.eps) / (1 - (1 - self.eps) * t)


class CosineNoise(Noise):
    def __init__(self, eps: float = 1e-3):
        self.eps = eps

    def total_noise(self, t):
        cos = torch.cos(t * torch.pi / 2)
        return -torch.log(self.eps + (1 - self.eps) * cos)

    def rate_noise(self, t):
        cos = (1 - self.eps) * torch.cos(t * torch.pi / 2)
        sin = (1 - self.eps) * torch.sin(t * torch.pi / 2)
        return (torch.pi / 2) * sin / (cos + self.eps)


def timestep_embedding(sigma: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=sigma.dtype) / half)
    args = sigma[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


@dataclass
class DDPMSchedule:
    num_steps: int
    timesteps: torch.Tensor   # [T+1]
    dt: float
    sigma_t: torch.Tensor     # [T+1]  (row T is the noise-removal sigma, model.py:576-577)
    mc_t: torch.Tensor        # [T+1]
    mc_s: torch.Tensor        # [T+1]
    t_freq: torch.Tensor      # [T+1, freq_dim] sinusoid of sigma_t


def ddpm_schedule(num_steps: int, eps: float = 1e-5, sample_max_t: float = 1.0, noise: Noise | None = None,
                  freq_dim: int = 256) -> DDPMSchedule:
    noise = noise or LogLinearNoise()
    ts = torch.linspace(sample_max_t, eps, num_steps + 1)
    dt = (1 - eps) / num_steps
    t = ts[:, None]
    sigma_t = noise(t)[0].squeeze(-1)
    sigma_s = noise(t - dt)[0].squeeze(-1)
    mc_t = 1 - torch.exp(-sigma_t)
    mc_s = 1 - torch.exp(-sigma_s)
    return DDPMSchedule(num_steps, ts, dt, sigma_t, mc_t, mc_s, timestep_embedding(sigma_t.float(), freq_dim))
