"""TEST INFRASTRUCTURE — torch-CPU float32 restatement of the reference's sampler layer.

Each function cites the reference lines it follows (paths relative to /root/reference).
Checked against tests/golden/g1..g8 (tests/test_oracle_golden.py); those fixtures were made by
running the reference's own code (tests/golden/make_goldens.py).
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Callable, List, Optional, Sequence

import torch

MASK = 4096            # C.STRUCTURE_MASK_TOKEN, slm/models/model.py:381
VOCAB = 4101           # slm/models/model.py:380
NEG_INF = -1000000.0   # slm/models/model.py:383


# ---- noise schedules: slm/utils/noise_utils.py -------------------------------------------------
class LogLinearNoiseRef:
    """noise_utils.py:188-213.  sigma(t) = -log1p(-(1-eps) t)."""

    def __init__(self, eps: float = 1e-3):
        self.eps = eps

    def total_noise(self, t: torch.Tensor) -> torch.Tensor:      # :205-206
        return -torch.log1p(-(1 - self.eps) * t)

    def rate_noise(self, t: torch.Tensor) -> torch.Tensor:       # :202-203
        return (1 - self.eps) / (1 - (1 - self.eps) * t)

    def __call__(self, t):                                       # Noise.forward :103-105
        return self.total_noise(t), self.rate_noise(t)


class CosineNoiseRef:
    """noise_utils.py:122-135 (the fallback of model.py:345-347)."""

    def __init__(self, eps: float = 1e-3):
        self.eps = eps

    def total_noise(self, t):
        return -torch.log(self.eps + (1 - self.eps) * torch.cos(t * torch.pi / 2))

    def rate_noise(self, t):
        c = (1 - self.eps) * torch.cos(t * torch.pi / 2)
        s = (1 - self.eps) * torch.sin(t * torch.pi / 2)
        return (torch.pi / 2) * s / (c + self.eps)

    def __call__(self, t):
        return self.total_noise(t), self.rate_noise(t)


class CosineSqrNoiseRef:
    """noise_utils.py:138-152.  sigma(t) = -log(eps + (1 - eps) cos^2(pi t / 2))."""

    def __init__(self, eps: float = 1e-3):
        self.eps = eps

    def total_noise(self, t):                                    # :149-151
        return -torch.log(self.eps + (1 - self.eps) * torch.cos(t * torch.pi / 2) ** 2)

    def rate_noise(self, t):                                     # :143-147
        c = (1 - self.eps) * (torch.cos(t * torch.pi / 2) ** 2)
        s = (1 - self.eps) * torch.sin(t * torch.pi)
        return (torch.pi / 2) * s / (c + self.eps)

    def __call__(self, t):
        return self.total_noise(t), self.rate_noise(t)


class LinearNoiseRef:
    """noise_utils.py:155-172.  sigma(t) = sigma_min + t (sigma_max - sigma_min); the rate is that constant (a 0-d tensor)."""

    def __init__(self, sigma_min=0, sigma_max=10, dtype=torch.float32):
        self.sigma_min = torch.tensor(sigma_min, dtype=dtype)
        self.sigma_max = torch.tensor(sigma_max, dtype=dtype)

    def total_noise(self, t):                                    # :164-165
        return self.sigma_min + t * (self.sigma_max - self.sigma_min)

    def rate_noise(self, t):                                     # :161-162
        return self.sigma_max - self.sigma_min

    def importance_sampling_transformation(self, t):             # :167-172
        f_T = torch.log1p(-torch.exp(-self.sigma_max))
        f_0 = torch.log1p(-torch.exp(-self.sigma_min))
        sigma_t = -torch.log1p(-torch.exp(t * f_T + (1 - t) * f_0))
        return (sigma_t - self.sigma_min) / (self.sigma_max - self.sigma_min)

    def __call__(self, t):
        return self.total_noise(t), self.rate_noise(t)


class GeometricNoiseRef:
    """noise_utils.py:175-185.  sigma(t) = sigma_min^(1-t) sigma_max^t."""

    def __init__(self, sigma_min=1e-3, sigma_max=1):
        self.sigmas = 1.0 * torch.tensor([sigma_min, sigma_max])

    def total_noise(self, t):                                    # :184-185
        return self.sigmas[0] ** (1 - t) * self.sigmas[1] ** t

    def rate_noise(self, t):                                     # :180-182
        return self.sigmas[0] ** (1 - t) * self.sigmas[1] ** t * (self.sigmas[1].log() - self.sigmas[0].log())

    def __call__(self, t):
        return self.total_noise(t), self.rate_noise(t)


def loglinear_importance_sampling_ref(t, eps: float = 1e-3):
    """LogLinearNoise.importance_sampling_transformation, noise_utils.py:208-213 (sigma_max / sigma_min as its __init__ sets
    them, :199-200)."""
    n = LogLinearNoiseRef(eps)
    sigma_max = n.total_noise(torch.tensor(1.0))
    sigma_min = eps + n.total_noise(torch.tensor(0.0))
    f_T = torch.log1p(-torch.exp(-sigma_max))
    f_0 = torch.log1p(-torch.exp(-sigma_min))
    sigma_t = -torch.log1p(-torch.exp(t * f_T + (1 - t) * f_0))
    return -torch.expm1(-sigma_t) / (1 - eps)


# ---- time conditioning: slm/models/net.py:486-522 ----------------------------------------------
def timestep_embedding_ref(sigma: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """net.py:497-517: cat[cos(sigma f), sin(sigma f)], f_k = exp(-ln(max_period) k / half)."""
    half = dim // 2
    k = torch.arange(start=0, end=half, dtype=sigma.dtype)
    freqs = torch.exp(-math.log(max_period) * k / half).to(dtype=sigma.dtype)
    ang = sigma[:, None] * freqs[None]
    out = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if dim % 2:
        out = torch.cat([out, torch.zeros_like(out[:, :1])], dim=-1)
    return out


class TimestepEmbedderRef(torch.nn.Module):
    """net.py:487-494,519-522: Linear(freq,h) -> SiLU -> Linear(h,h) on the sinusoid of sigma."""

    def __init__(self, hidden: int, freq: int = 256):
        super().__init__()
        self.mlp = torch.nn.Sequential(torch.nn.Linear(freq, hidden), torch.nn.SiLU(),
                                       torch.nn.Linear(hidden, hidden))
        self.freq = freq

    def forward(self, sigma):
        return self.mlp(timestep_embedding_ref(sigma, self.freq))


# ---- the sampler: slm/models/model.py -----------------------------------------------------------
def logits_parameterization_ref(logits: torch.Tensor, xt: torch.Tensor) -> torch.Tensor:
    """model.py:527-533 (works on a copy; the reference mutates the network output)."""
    z = logits.clone()
    z[:, :, MASK] += NEG_INF
    z = z - torch.logsumexp(z, dim=-1, keepdim=True)
    known = xt != MASK
    z[known] = NEG_INF
    z[known, xt[known]] = 0
    return z


def sample_categorical_ref(q: torch.Tensor, u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """model.py:24-28.  `u` None draws torch.rand_like(q) exactly like the reference."""
    if u is None:
        u = torch.rand_like(q)
    g = 1e-10 - (u + 1e-10).log()
    return (q / g).argmax(dim=-1)


class MDLMSamplerRef:
    """The inference half of MaskedDiffusionLanguageModeling (model.py:464-492, 535-607)."""

    def __init__(self, net: Callable, sigma_embedder: Optional[Callable], noise=None,
                 time_conditioning: bool = True, noise_removal: bool = True):
        self.net = net
        self.sigma_embedder = sigma_embedder
        self.noise = noise if noise is not None else CosineNoiseRef(1e-3)   # model.py:345-347
        self.time_conditioning = time_conditioning
        self.noise_removal = noise_removal

    def model_wrapper(self, xt, sequence_tokens, sigma):                    # model.py:464-492
        cond = None
        if sigma is not None:
            if sigma.ndim > 1:                                               # _process_sigma :535-541
                sigma = sigma.squeeze(-1)
            if not self.time_conditioning:
                sigma = torch.zeros_like(sigma)
            cond = self.sigma_embedder(sigma.to(torch.float32))
            cond = torch.tile(cond[:, None, :], (1, xt.shape[1], 1))
        out = self.net(structure_tokens=xt, sequence_tokens=sequence_tokens, auxiliary_embeddings=cond,
                       labels=None)
        return logits_parameterization_ref(out.structure_logits, xt)

    def ddpm_update(self, x, t, sequence_tokens, dt, u=None, return_logp=False):   # model.py:583-607
        sigma_t = self.noise(t)[0].squeeze(-1)
        sigma_s = self.noise(t - dt)[0].squeeze(-1)
        mc_t = (1 - torch.exp(-sigma_t))[:, None, None]
        mc_s = (1 - torch.exp(-sigma_s))[:, None, None]
        log_p = self.model_wrapper(x, sequence_tokens, sigma_t)
        q = log_p.exp() * (mc_t - mc_s)
        q[:, :, MASK] = mc_s[:, :, 0]
        drawn = sample_categorical_ref(q, u)
        keep = (x != MASK).to(x.dtype)
        new = keep * x + (1 - keep) * drawn
        return (new, log_p) if return_logp else new

    @torch.no_grad()
    def ddpm_sample(self, sequence_tokens, num_steps, eps=1e-5, input_prior=None, sample_max_t=1.0,
                    trajectory: Optional[list] = None):                     # model.py:543-581
        if input_prior is None:
            x = torch.full(tuple(sequence_tokens.shape), MASK, dtype=torch.int64)
            assert sample_max_t == 1.0
        else:
            x = input_prior.clone()
            assert x.shape == sequence_tokens.shape
        timesteps = torch.linspace(sample_max_t, eps, num_steps + 1)
        dt = (1 - eps) / num_steps
        for i in range(num_steps):
            t = timesteps[i] * torch.ones(x.shape[0], 1)
            x = self.ddpm_update(x, t, sequence_tokens, dt)
            if trajectory is not None:
                trajectory.append(x.clone())
        if self.noise_removal:
            t = timesteps[-1] * torch.ones(x.shape[0], 1)
            sigma = self.noise(t)[0]
            x = self.model_wrapper(x, sequence_tokens, sigma).argmax(dim=-1)
        return x


def ddpm_schedule_ref(num_steps: int, eps: float = 1e-5, sample_max_t: float = 1.0, noise=None):
    """Per-step scalars exactly as model.py:564-567,584-595 produces them (float32 tensors)."""
    noise = noise or LogLinearNoiseRef()
    ts = torch.linspace(sample_max_t, eps, num_steps + 1)
    dt = (1 - eps) / num_steps
    t = ts[:, None]
    sig_t = noise(t)[0].squeeze(-1)
    sig_s = noise(t - dt)[0].squeeze(-1)
    return {"timesteps": ts, "dt": dt, "sigma_t": sig_t, "sigma_s": sig_s,
            "mc_t": 1 - torch.exp(-sig_t), "mc_s": 1 - torch.exp(-sig_s)}


# ---- host driver pieces: slm/sample_esmdiff.py --------------------------------------------------
def batch_split_ref(n_tokens_or_residues: int, num_samples: int, n_max_residue_square: int = 200 * 200 * 105
                    ) -> List[int]:
    """sample_esmdiff.py:181-193 (ddpm: length in tokens) and :104-112 (gibbs: len(protseq))."""
    sq = n_tokens_or_residues * n_tokens_or_residues
    total = sq * num_samples
    sizes = [n_max_residue_square // sq] * (total // n_max_residue_square)
    if total % n_max_residue_square > 0:
        sizes.append(num_samples - sum(sizes))
    return sizes


def merge_pdbfiles_ref(pdb_files: Sequence[Path], save_to: Path) -> None:
    """slm/utils/eval_utils.py:437-492: MODEL n / ATOM+TER lines / ENDMDL, 80-col padded, final END."""
    out: List[str] = []
    n_model = 0
    for f in pdb_files:
        lines = Path(f).read_text().splitlines(keepends=True)
        multi = any(ln.startswith("MODEL") or ln.startswith("ENDMDL") for ln in lines)
        if not multi:
            n_model += 1
            out.append(f"MODEL     {n_model}")
            out += [ln.strip() for ln in lines if ln.startswith("TER") or ln.startswith("ATOM")]
            out.append("ENDMDL")
        else:
            for ln in lines:
                if ln.startswith("MODEL"):
                    n_model += 1
                    if n_model > 1:
                        out.append("ENDMDL")
                    out.append(f"MODEL     {n_model}")
                elif ln.startswith("END"):
                    continue
                elif ln.startswith("TER") or ln.startswith("ATOM"):
                    out.append(ln.strip())
    out.append("ENDMDL")
    out.append("END")
    Path(save_to).parent.mkdir(parents=True, exist_ok=True)
    Path(save_to).write_text("\n".join(ln.ljust(80) for ln in out) + "\n")
