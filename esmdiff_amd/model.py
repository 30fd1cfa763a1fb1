"""Drop-in for the inference surface of the reference's task module
(/root/reference/slm/models/model.py:316-607, class MaskedDiffusionLanguageModeling) on top of the HIP engine.

Kept: constructor flags that matter at inference (time_conditioning, noise_removal, noise schedule), the
attributes the CLI touches (`.net`, `.noise_removal`, `.device`), and
    ddpm_sample(sequence_tokens, num_steps=None, eps=1e-5, input_prior=None, sample_max_t=1.0) -> (B, L) int64
with the reference's argument meaning and error behaviour (model.py:543-581).
Added (keyword-only): `seed`, `sample_offset` (Philox noise, sharding independent) and `noise="torch-cpu"`,
which draws the uniforms exactly like the reference's CPU path — torch.rand_like on the CPU generator, one
(B, L, 4101) tensor per update in the same order (model.py:27) — and uploads them (parity mode).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional

import torch

from .config import ESM3_OPEN, ModelConfig
from .constants import STRUCTURE_MASK_TOKEN, STRUCTURE_VOCAB
from .engine import Engine
from .schedule import CosineNoise, CosineSqrNoise, GeometricNoise, Linear, LogLinearNoise, Noise, ddpm_schedule
from .weights import load_checkpoint_state_dict, random_init_state_dict


class MaskedDiffusionLanguageModeling:
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: ModelConfig = ESM3_OPEN,
                 noise_schedule: Optional[Noise] = None, max_batch: int = 128, max_len: int = 1026,
                 device: int = 0, noise_removal: bool = True, precision: str = "bf16", step0_sharing: bool = True,
                 head_precision: Optional[str] = None):
        if noise_schedule is None:
            print("Using default noise schedule: CosineNoise(eps=1e-3)")    # model.py:345-347
            noise_schedule = CosineNoise(eps=1e-3)
        self.cfg = cfg
        self.noise = noise_schedule
        self.time_conditioning = cfg.time_conditioning
        self.noise_removal = noise_removal
        self.vocab_size = STRUCTURE_VOCAB
        self.mask_index = STRUCTURE_MASK_TOKEN
        self.neg_infinity = -1000000.0
        # precision="certified": the f32-grade engine is `net` (every generic path — parity noise, _model_wrapper — runs on it);
        # the Philox ddpm loop and the gibbs loop (esmdiff_amd.gibbs.iterative_sampling_raw) draw on an f16 engine and verify only
        # the decisions its measured error leaves open on `net` (certified.py): the ids of `net`'s chain at about 2.4x its rate
        self.certified = None
        if precision == "certified":
            from .certified import CertifiedSampler
            self.net = Engine(cfg, state_dict, max_batch=max_batch, max_len=max_len, device=device, precision="f32_split")
            self.fast = Engine(cfg, state_dict, max_batch=max_batch, max_len=max_len, device=device, precision="f16",
                               head_precision=head_precision or "f32")    # f32-grade head: 40 % less logit error, 1/3 fewer re-runs
            self.certified = CertifiedSampler(self.fast, self.net)
        else:
            self.net = Engine(cfg, state_dict, max_batch=max_batch, max_len=max_len, device=device, precision=precision,
                              head_precision=head_precision)
        # exact step-0 sharing (include/esmdiff_hip.h): the CLI repeats ONE sequence per batch, so the first forward of a run has
        # identical rows; the engine checks that on the device and then serves all samples from a sub-batch forward — ids are
        # bit-identical to the unshared loop (tests/test_gpu_fullwidth.py::test_step0_sharing_is_exact).  bench.py's headline
        # run builds its Engine directly and leaves it off.
        self.net.set_step0_sharing(step0_sharing)
        self.net.set_final_skip(step0_sharing)      # the other exact shortcut: noise-removal forward only for samples with a MASK left
        self.device = self.net.device
        self._parity_gen = None      # noise="torch-cpu": ONE generator stream per run, like the reference's global RNG
        self._parity_seed = None

    def eval(self):
        return self

    def to(self, device):
        return self

    def reset_parity_stream(self, seed: int) -> None:
        """noise="torch-cpu": restart the replay of the reference's process-wide torch.rand stream at `seed` — what a
        torch.manual_seed(seed) before the reference's first ddpm_sample does (the reference itself never re-seeds, so
        every later batch and target runs on in the same stream).  Implicit at the first parity call of a model and when
        the seed changes; tests that replay the reference's goldens call it where the golden script seeded."""
        self._parity_gen = torch.Generator().manual_seed(seed)         # same mt19937 stream as torch.manual_seed(seed)
        self._parity_seed = seed

    def _sample_prior(self, *batch_dims):
        return self.mask_index * torch.ones(*batch_dims, dtype=torch.int64)

    sequence_prediction = False      # the reference's flag (model.py:332, 366, 487-490): also return the sequence head's logits; needs
                                     # output_heads.sequence_head.* in the state dict (net.py:299-311), as the reference asserts (:374-375)

    def logits_parameterization(self, logits: torch.Tensor, xt: torch.Tensor) -> torch.Tensor:
        """model.py:527-533 on the device, in the reference's operation order (torch ops on the logits tensor: the loops use the
        fused sampler kernel instead; this is the standalone form `_model_wrapper` returns)."""
        logits = logits.clone()
        logits[:, :, self.mask_index] += self.neg_infinity
        logits = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
        unmasked = xt != self.mask_index
        logits[unmasked] = self.neg_infinity
        logits[unmasked, xt[unmasked]] = 0
        return logits

    @torch.no_grad()
    def _model_wrapper(self, xt, sequence_tokens=None, sigma=None, shield_special_tokens: bool = False):
        """The reference's `_model_wrapper` (model.py:464-492): log-probabilities (B, L, 4101) of the SUBS parameterisation at
        noise level `sigma` ((B,) or (B, 1): one value per sample as in the reference, or a single value; None = no time
        conditioning at all), optionally with the five special ids shielded (:484-486).
        Returns (logits, None) like the reference with sequence_prediction off, (logits, sequence_logits (B, L, n_sequence_heads))
        with it on (:488-490: the raw output of the network's sequence head for the same forward)."""
        from .schedule import timestep_embedding
        if self.sequence_prediction and not getattr(self.net, "n_sequence_heads", 0):
            raise AssertionError("Sequence head not found in tbe network, but sequence_prediction is True.")      # model.py:375
        xt = xt.to(self.device)
        B, L = xt.shape
        if sequence_tokens is None:     # net.py:412-416: the sequence track defaults to all-mask
            from .constants import SEQUENCE_MASK_TOKEN
            sequence_tokens = torch.full((B, L), SEQUENCE_MASK_TOKEN, dtype=torch.int64)
        tf = None
        if sigma is not None:
            sg = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1)          # _process_sigma: squeeze
            if sg.numel() not in (1, B):
                raise ValueError(f"_model_wrapper: sigma must hold 1 or B = {B} values, got {sg.numel()}")
            if bool((sg != sg[0]).any()):     # one sigma per sample (model.py:466-471): esmdiff_forward_logits_sigmas
                tf = self.net.conditioning_rows(timestep_embedding(sg, self.cfg.freq_dim))
            else:                             # what the sampling loop passes: one sigma for the whole batch
                tf = self.net.conditioning_rows(timestep_embedding(sg[:1], self.cfg.freq_dim))
                tf = None if tf is None else tf[0]
        raw = self.net.forward_logits(xt, sequence_tokens.to(self.device), tf)
        logits = self.logits_parameterization(raw.float(), xt)
        if shield_special_tokens:
            logits[..., STRUCTURE_MASK_TOKEN:STRUCTURE_MASK_TOKEN + 5] += self.neg_infinity
        if self.sequence_prediction:
            return logits, self.net.sequence_logits(B, L)
        return logits, None

    def _process_sigma(self, sigma):
        """model.py:535-541: (B, 1) -> (B,); zeros when the model is not time-conditioned."""
        if sigma.ndim > 1:
            sigma = sigma.squeeze(-1)
        if not self.time_conditioning:
            sigma = torch.zeros_like(sigma)
        assert sigma.ndim == 1, sigma.shape
        return sigma

    @torch.no_grad()
    def _ddpm_update(self, x, t, sequence_tokens, dt, *, u=None, seed: Optional[int] = None, sample_offset: int = 0, step: int = 0,
                     t_freq: Optional[torch.Tensor] = None):
        """One reverse-diffusion update, the reference's `_ddpm_update(x, t, sequence_tokens, dt)` (model.py:583-607): t is (B, 1);
        sigma_t = noise(t), sigma_s = noise(t - dt), move chances 1 - exp(-sigma); network at sigma_t; draw; carry the unmasked
        rows over.  Noise: explicit uniforms `u` (B, L, 4101) — what torch.rand_like(q_xs) draws — or Philox(seed, sample, step).
        x is updated in place and returned.  Rows of t may differ (the reference allows it; its loop never does it): the network
        then runs with one sigma per sample and the draw runs per sample.  t_freq: the conditioning row(s) already prepared
        for sigma_t by the caller (ddpm_sample passes its schedule table's row), else computed here."""
        from .schedule import timestep_embedding
        x = x.to(self.device).contiguous()
        seq = sequence_tokens.to(self.device)
        B, L = x.shape
        t = torch.as_tensor(t, dtype=torch.float32).reshape(B, 1).cpu()
        sigma_t = self.noise(t)[0].squeeze(-1)
        sigma_s = self.noise(t - dt)[0].squeeze(-1)
        mc_t, mc_s = 1 - torch.exp(-sigma_t), 1 - torch.exp(-sigma_s)
        shared = bool((t == t[0]).all())
        if t_freq is not None and not shared and torch.as_tensor(t_freq).dim() != 2:
            # one conditioning row for samples at different noise levels: the network would see one sigma, the draws another
            raise ValueError("_ddpm_update: t differs between samples, so t_freq must hold one row per sample (B, freq_dim) or be None")
        if t_freq is None:
            tf = self.net.conditioning_rows(timestep_embedding(sigma_t[:1] if shared else sigma_t, self.cfg.freq_dim))
            t_freq = None if tf is None else (tf[0] if shared else tf)
        logits = self.net.forward_logits(x, seq, t_freq)
        if u is None and seed is None:
            raise ValueError("_ddpm_update needs explicit uniforms `u` or a Philox `seed`")
        if shared:
            self.net.ddpm_step(x, logits, float(mc_t[0]), float(mc_s[0]), u=u, seed=None if u is not None else seed,
                               sample_offset=sample_offset, step=step)
        else:
            for b in range(B):
                self.net.ddpm_step(x[b:b + 1], logits[b:b + 1], float(mc_t[b]), float(mc_s[b]), u=None if u is None else u[b:b + 1],
                                   seed=None if u is not None else seed, sample_offset=sample_offset + b, step=step)
        return x

    @torch.no_grad()
    def ddpm_sample(self, sequence_tokens, num_steps=None, eps=1e-5, input_prior=None, sample_max_t=1.0, *,
                    seed: int = 0, sample_offset: int = 0, noise: str = "philox"):
        if num_steps is None:
            print("Using by default num_steps: 1000")
            num_steps = 1000
        if input_prior is None:
            assert sample_max_t == 1.0, "sample_max_t has to be 1.0 when input_prior is None"
        else:
            print(f"Using input_prior: {input_prior.shape}")
            assert tuple(input_prior.shape) == tuple(sequence_tokens.shape), \
                f"Invalid input_prior shape: {input_prior.shape} v.s. (seq) {sequence_tokens.shape}"
        sch = ddpm_schedule(num_steps, eps, sample_max_t, self.noise, self.cfg.freq_dim)
        B, L = sequence_tokens.shape
        if noise == "philox" and self.noise_removal:
            loop = self.certified if self.certified is not None else self.net
            return loop.ddpm_sample(sequence_tokens, sch, seed=seed, sample_offset=sample_offset, input_prior=input_prior)
        # step-by-step drive (parity mode / noise_removal off)
        seq = sequence_tokens.to(self.device)
        x = (self._sample_prior(B, L) if input_prior is None else input_prior.clone()).to(self.device).contiguous()
        tf = self.net.conditioning_rows(sch.t_freq)
        tf = [None] * (num_steps + 1) if tf is None else tf
        if noise == "torch-cpu":
            # The reference never seeds inside its sampling path: torch.rand_like keeps consuming the ONE process-wide CPU
            # stream (model.py:27) across batches AND across targets, from wherever the caller's torch.manual_seed left it.  The
            # private generator below is that stream: created at the first parity call of a run (or when the seed
            # changes, or after reset_parity_stream()) and never restarted per target or per batch.
            if self._parity_gen is None or self._parity_seed != seed:
                self.reset_parity_stream(seed)
            gen = self._parity_gen
        elif noise != "philox":
            raise ValueError(f"unknown noise source {noise!r}")
        for i in range(num_steps):
            t = sch.timesteps[i] * torch.ones(B, 1)
            u = torch.rand(B, L, STRUCTURE_VOCAB, generator=gen) if noise == "torch-cpu" else None   # == torch.rand_like(q_xs), model.py:27
            x = self._ddpm_update(x, t, seq, sch.dt, u=u, seed=seed, sample_offset=sample_offset, step=i, t_freq=tf[i])
        if self.noise_removal:
            logits = self.net.forward_logits(x, seq, tf[num_steps])
            self.net.ddpm_step(x, logits, 0.0, 0.0, final=True)
        return x


def config_from_hydra_yaml(path, cfg: ModelConfig = ESM3_OPEN):
    """The inference-time content of a run's `.hydra/config.yaml` (checkpoint_utils.py:45-57 instantiates `cfg.model`
    from it): noise schedule, time_conditioning, structure-head width.  Returns (ModelConfig, Noise)."""
    import dataclasses

    import yaml
    m = (yaml.safe_load(Path(path).read_text()) or {}).get("model", {}) or {}
    ns = m.get("noise_schedule") or {}
    target = str(ns.get("_target_", "slm.utils.noise_utils.LogLinearNoise")).rsplit(".", 1)[-1]
    # every schedule class of noise_utils.py:122-213 (hydra passes the yaml's keys to the constructor)
    kinds = {"LogLinearNoise": (LogLinearNoise, ("eps",)), "CosineNoise": (CosineNoise, ("eps",)),
             "CosineSqrNoise": (CosineSqrNoise, ("eps",)), "Linear": (Linear, ("sigma_min", "sigma_max")),
             "GeometricNoise": (GeometricNoise, ("sigma_min", "sigma_max"))}
    if target not in kinds:
        raise NotImplementedError(f"noise schedule {target} (from {path}) is not one of {sorted(kinds)}")
    cls, keys = kinds[target]
    noise = cls(**{k: float(ns[k]) for k in keys if k in ns})
    net = m.get("net") or {}
    cfg = dataclasses.replace(cfg, time_conditioning=bool(m.get("time_conditioning", cfg.time_conditioning)),
                              n_structure_heads=int(net.get("n_structure_heads", cfg.n_structure_heads)))
    return cfg, noise


def load_state_dict_from_lightning_ckpt(ckpt_path, device="cuda", max_batch: int = 128, max_len: int = 1026,
                                        cfg: ModelConfig = ESM3_OPEN, precision: str = "bf16",
                                        head_precision: Optional[str] = None):
    """/root/reference/slm/utils/checkpoint_utils.py:41-74: a `.pt` whose 'module' dict holds `net.*` and
    `sigma_embedder.*`; the model is what the run's `.hydra/config.yaml` says when that file sits where the reference
    looks for it (:45-50), else the mdlm.yaml configuration (LogLinearNoise, time conditioning, 4101-way head);
    noise_removal is forced on (:71).  Under torch.distributed this is a collective (dist.broadcast_state_dict): every rank
    must call it; a loading error on the reading rank is raised on all of them."""
    from .weights import checkpoint_file_and_config
    print(f"Loading ESMDiff ckpt from {ckpt_path}")
    _, exp_cfg_path = checkpoint_file_and_config(ckpt_path)
    noise = LogLinearNoise()
    if exp_cfg_path is not None:
        cfg, noise = config_from_hydra_yaml(exp_cfg_path, cfg)
    else:
        print("Config file not found next to the checkpoint. Use default config (configs/experiment/mdlm.yaml).")
    print(f"Loaded experiment config: {exp_cfg_path or 'mdlm.yaml defaults'}...")
    # one process per GPU: rank 0 reads the file, the others get the tensors by one broadcast (dist.broadcast_state_dict)
    from .dist import broadcast_state_dict
    dev = torch.device(device).index or 0
    sd, load_t = broadcast_state_dict(lambda: load_checkpoint_state_dict(ckpt_path), torch.device("cuda", dev))
    model = MaskedDiffusionLanguageModeling(sd, cfg, noise, max_batch, max_len, dev, noise_removal=True, precision=precision,
                                            head_precision=head_precision)
    model.load_timings = load_t
    del sd
    print(f"Sucessfully loaded model from {ckpt_path}... (rank {load_t['rank']}: {load_t['load_s']} s, {load_t['bytes'] / 1e9:.2f} GB, "
          f"{load_t['path']})")
    return model


def load_stock_esm3(path, device="cuda", max_batch: int = 128, max_len: int = 1026, precision: str = "bf16",
                    head_precision: Optional[str] = None):
    """The pre-trained ESM3 the reference uses when --ckpt is absent (`ESM3.from_pretrained("esm3_sm_open_v1")`,
    /root/reference/slm/sample_esmdiff.py:37, :252-255; gibbs mode only): a plain esm state dict (keys without the
    `net.` prefix, 4096-way structure head, no sigma_embedder) saved with torch.save."""
    from .config import ESM3_OPEN_STOCK
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    dev = torch.device(device).index or 0
    return MaskedDiffusionLanguageModeling(sd, ESM3_OPEN_STOCK, None, max_batch, max_len, dev, noise_removal=True,
                                           precision=precision, head_precision=head_precision)


def random_init_model(cfg: ModelConfig = ESM3_OPEN, seed: int = 0, max_batch: int = 128, max_len: int = 1026,
                      device: int = 0, precision: str = "bf16", head_precision: Optional[str] = None):
    """ESM3-open-sized random weights (no checkpoint can be fetched offline): synthetic benchmarking / tests."""
    sd = random_init_state_dict(cfg, seed=seed, device=f"cuda:{device}", with_geom=True)   # real checkpoints carry geom_attn too
    return MaskedDiffusionLanguageModeling(sd, cfg, LogLinearNoise(), max_batch, max_len, device, noise_removal=True,
                                           precision=precision, head_precision=head_precision)
