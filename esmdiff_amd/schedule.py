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
        self.sigma_max = self.total_noise(torch.tensor(1.0))                # noise_utils.py:199-200
        self.sigma_min = self.eps + self.total_noise(torch.tensor(0.0))

    def importance_sampling_transformation(self, t):                        # noise_utils.py:208-213
        f_T = torch.log1p(-torch.exp(-self.sigma_max))
        f_0 = torch.log1p(-torch.exp(-self.sigma_min))
        sigma_t = -torch.log1p(-torch.exp(t * f_T + (1 - t) * f_0))
        return -torch.expm1(-sigma_t) / (1 - self.eps)

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


class CosineSqrNoise(Noise):
    """noise_utils.py:138-152."""

    def __init__(self, eps: float = 1e-3):
        self.eps = eps

    def total_noise(self, t):
        cos2 = torch.cos(t * torch.pi / 2) ** 2
        return -torch.log(self.eps + (1 - self.eps) * cos2)

    def rate_noise(self, t):
        cos2 = (1 - self.eps) * (torch.cos(t * torch.pi / 2) ** 2)
        sin = (1 - self.eps) * torch.sin(t * torch.pi)
        return (torch.pi / 2) * sin / (cos2 + self.eps)


class Linear(Noise):
    """noise_utils.py:155-172: sigma grows linearly from sigma_min to sigma_max."""

    def __init__(self, sigma_min=0, sigma_max=10, dtype=torch.float32):
        self.sigma_min = torch.tensor(sigma_min, dtype=dtype)
        self.sigma_max = torch.tensor(sigma_max, dtype=dtype)

    def total_noise(self, t):
        return self.sigma_min + t * (self.sigma_max - self.sigma_min)

    def rate_noise(self, t):
        return self.sigma_max - self.sigma_min

    def importance_sampling_transformation(self, t):
        f_T = torch.log1p(-torch.exp(-self.sigma_max))
        f_0 = torch.log1p(-torch.exp(-self.sigma_min))
        sigma_t = -torch.log1p(-torch.exp(t * f_T + (1 - t) * f_0))
        return (sigma_t - self.sigma_min) / (self.sigma_max - self.sigma_min)


class GeometricNoise(Noise):
    """noise_utils.py:175-185: sigma(t) = sigma_min^(1-t) * sigma_max^t."""

    def __init__(self, sigma_min=1e-3, sigma_max=1):
        self.sigmas = 1.0 * torch.tensor([sigma_min, sigma_max])

    def total_noise(self, t):
        return self.sigmas[0] ** (1 - t) * self.sigmas[1] ** t

    def rate_noise(self, t):
        return self.total_noise(t) * (self.sigmas[1].log() - self.sigmas[0].log())


def get_noise(noise_type: str, sigma_min=None, sigma_max=None, dtype=torch.float32) -> Noise:
    """`get_noise(config)` (noise_utils.py:75-90) with the config fields as arguments: type in geometric / loglinear / cosine /
    cosinesqr / linear; sigma_min / sigma_max for geometric and linear."""
    if noise_type == "geometric":
        return GeometricNoise(sigma_min, sigma_max)
    if noise_type == "loglinear":
        return LogLinearNoise()
    if noise_type == "cosine":
        return CosineNoise()
    if noise_type == "cosinesqr":
        return CosineSqrNoise()
    if noise_type == "linear":
        return Linear(sigma_min, sigma_max, dtype)
    raise ValueError(f"{noise_type} is not a valid noise")


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
