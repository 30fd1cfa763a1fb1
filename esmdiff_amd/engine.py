"""Python host side of the C ABI (include/esmdiff_hip.h): owns an esmdiff_engine and exposes the call
sites of the reference's hot path with PyTorch-ROCm tensors as plain device-memory containers.

  Engine.forward_logits  <- self.net(...).structure_logits           model.py:475-481
  Engine.ddpm_step       <- _ddpm_update's sampling half              model.py:583-607, 24-28
  Engine.ddpm_sample     <- MaskedDiffusionLanguageModeling.ddpm_sample  model.py:543-581
(paths relative to /root/reference/slm/models)
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _native as N
from .config import ModelConfig
from .constants import STRUCTURE_MASK_TOKEN, STRUCTURE_VOCAB
from .schedule import DDPMSchedule


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("esmdiff_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False and "
                           "there is no CPU fallback")


class Engine:
    """One engine per (process, device).  Not thread-safe.  All work is enqueued on torch's current stream."""

    def __init__(self, cfg: ModelConfig, state_dict: Dict[str, torch.Tensor], max_batch: int, max_len: int,
                 device: int = 0, precision: str = "bf16", head_precision: Optional[str] = None):
        """precision: "bf16" (the throughput path: bf16 MFMA, f32 accumulate and residual stream), "f32" (the strict
        path, csrc/strict.hip: float32 weights and activations on the f32-input MFMA — the reference's own arithmetic,
        checkpoint_utils.py:59-73; ~1/12 of the throughput; the referee) or "f32_split" (the strict path with every large
        linear as three f16 MFMA passes over split operands, csrc/gemm_split.hip: float32-grade, ~1/4 of the bf16 throughput) or
        "f16" (the bf16 path's kernels compiled with IEEE-half operands, csrc/ed_half.h: same speed, 1/8 of the operand rounding).
        head_precision="f32" (bf16 / f16 engines): final LayerNorm + output head in float32 grade on the split kernels."""
        _require_gpu()
        if precision not in N.PRECISION:
            raise ValueError(f"precision must be one of {sorted(N.PRECISION)}, got {precision!r}")
        if head_precision not in (None, "f32") and head_precision != precision:
            raise ValueError(f"head_precision must be None / {precision!r} (the body's precision) or 'f32', got {head_precision!r}")
        self.precision = precision
        self.head_precision = "f32" if (head_precision == "f32" or precision in ("f32", "f32_split")) else precision
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.max_batch, self.max_len = max_batch, max_len
        self._lib = N.lib()
        self._h = ctypes.c_void_p(0)
        c = N.Config(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.ffn_hidden, cfg.n_structure_heads, cfg.freq_dim,
                     max_batch, max_len, cfg.residue_scale, int(cfg.time_conditioning), N.PRECISION[precision],
                     1 if (head_precision == "f32" and precision in ("bf16", "f16")) else 0)
        # upload the caller's tensors (any float dtype) as device containers; the engine makes its own
        # bf16 / re-laid-out copies, after which these are released.
        keep, table = [], (N.Weight * len(state_dict))()
        with torch.cuda.device(self.device):
            for i, (name, t) in enumerate(state_dict.items()):
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                d = t.detach().to(self.device).contiguous()
                keep.append(d)
                shape = (ctypes.c_int64 * 4)(*(list(d.shape) + [0] * (4 - d.dim())))
                table[i] = N.Weight(name.encode(), d.data_ptr(), N.DT_F32 if d.dtype == torch.float32 else N.DT_BF16,
                                    d.dim(), shape)
            torch.cuda.synchronize()
            code = self._lib.esmdiff_engine_create(ctypes.byref(c), table, len(state_dict), device,
                                                   ctypes.byref(self._h))
        if code != 0:
            raise RuntimeError(f"esmdiff_engine_create failed ({code}): "
                               f"{self._lib.esmdiff_last_error(None).decode()}")
        del keep
        self.ld_logits = (cfg.n_structure_heads + 3) // 4 * 4
        self.has_geom = any(k.endswith("transformer.blocks.0.geom_attn.proj.weight") for k in state_dict)
        self.has_sigma_embedder = any(k.startswith("sigma_embedder.mlp.0.") for k in state_dict)
        self.n_sequence_heads = next((int(v.shape[0]) for k, v in state_dict.items() if k.endswith("output_heads.sequence_head.3.weight")), 0)
        self._frames = None

    def branch_linear_layernorm(self, A: torch.Tensor, W: torch.Tensor, x: torch.Tensor, alpha: float, w: torch.Tensor,
                                b: Optional[torch.Tensor]):
        """x += alpha * (A @ W^T) in place (f32), returns (LayerNorm(x) * w (+ b) as bf16, K slices used): the small-batch
        form of a residual branch (esmdiff_branch_linear_layernorm)."""
        M, K = A.shape
        Nn = W.shape[0]
        assert A.dtype == W.dtype == torch.bfloat16 and x.dtype == torch.float32 and x.shape == (M, Nn)
        assert A.is_contiguous() and W.is_contiguous() and x.is_contiguous()
        y = torch.empty(M, Nn, dtype=torch.bfloat16, device=A.device)
        S = ctypes.c_int32(0)
        self._chk(self._lib.esmdiff_branch_linear_layernorm(self._h, _ptr(A), _ptr(W), _ptr(x), float(alpha),
                                                            _ptr(w.float().contiguous()),
                                                            _ptr(None if b is None else b.float().contiguous()), _ptr(y),
                                                            M, Nn, K, ctypes.byref(S), _stream()))
        return y, int(S.value)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.esmdiff_engine_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def _chk(self, code):
        N.check(code, self._h)

    def _tok(self, t: torch.Tensor, B: int, L: int) -> torch.Tensor:
        t = t.to(device=self.device, dtype=torch.int64)
        if t.dim() == 1:
            t = t[None].expand(B, L)
        return t.contiguous()

    def _check_ids(self, seq: torch.Tensor, x: torch.Tensor) -> None:
        """The embedding tables hold 64 sequence rows and 4101 structure rows (net.py:445-466; -1 means MASK): an id
        outside them is a caller error, reported here instead of being looked up."""
        bad = ((seq < 0) | (seq > 63)).any() | ((x < -1) | (x >= STRUCTURE_VOCAB)).any()
        if bool(bad):
            raise ValueError(f"token id out of range: sequence ids must be in 0..63 (got {int(seq.min())}..{int(seq.max())}), "
                             f"structure ids in -1..{STRUCTURE_VOCAB - 1} (got {int(x.min())}..{int(x.max())})")

    def conditioning_rows(self, t_freq: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """What `_model_wrapper` conditions on (model.py:464-471): sigma_embedder(sigma) with time conditioning,
        sigma_embedder(0) without it when the embedder exists (_process_sigma zeroes sigma, model.py:535-541), nothing
        for a model without a sigma embedder.  In: (n, freq_dim) sinusoids of the real sigmas; out: the rows to feed."""
        if t_freq is None or not self.has_sigma_embedder:
            return None
        if self.cfg.time_conditioning:
            return t_freq
        from .schedule import timestep_embedding
        zero = timestep_embedding(torch.zeros(1, dtype=torch.float32), self.cfg.freq_dim)
        return zero.expand(t_freq.shape[0], -1).contiguous() if t_freq.dim() == 2 else zero[0]

    def forward_logits(self, x: torch.Tensor, sequence_tokens: torch.Tensor, t_freq: Optional[torch.Tensor],
                       out: Optional[torch.Tensor] = None, check_ids: bool = True) -> torch.Tensor:
        """x, sequence_tokens: (B,L) int64.  t_freq: (freq_dim,) f32 sinusoid of the sigma all samples share, (B, freq_dim) for
        one sigma per sample (esmdiff_forward_logits_sigmas), or None.
        Returns raw structure logits, a (B,L,4101) view of a (B,L,ld) float32 buffer."""
        B, L = x.shape
        x = self._tok(x, B, L)
        seq = self._tok(sequence_tokens, B, L)
        if check_ids:      # (a device -> host read-back: loops that feed the engine its own output skip it after the first call)
            self._check_ids(seq, x)
        if out is None:
            out = torch.empty(B, L, self.ld_logits, dtype=torch.float32, device=self.device)
        tf = None if t_freq is None else t_freq.to(device=self.device, dtype=torch.float32).contiguous()
        if tf is not None and tf.dim() == 2:
            if tf.shape[0] != B:
                raise ValueError(f"per-sample conditioning: {tf.shape[0]} sinusoid rows for a batch of {B}")
            fn = self._lib.esmdiff_forward_logits_sigmas
        else:
            fn = self._lib.esmdiff_forward_logits
        self._chk(fn(self._h, _ptr(seq), _ptr(x), _ptr(tf), _ptr(out), out.shape[-1], B, L, _stream()))
        return out[..., :self.cfg.n_structure_heads]

    def embeddings(self, B: int, L: int) -> torch.Tensor:
        """ESMOutput.embeddings of the forward that just ran (net.py:468-469): the pre-norm hidden state, (B, L, d) f32."""
        out = torch.empty(B, L, self.cfg.d_model, dtype=torch.float32, device=self.device)
        self._chk(self._lib.esmdiff_get_embeddings(self._h, _ptr(out), B, L, _stream()))
        return out

    def sequence_logits(self, B: int, L: int) -> torch.Tensor:
        """ESMOutput.sequence_logits of the forward that just ran (net.py:310-311; esmdiff_get_sequence_logits): (B, L,
        n_sequence_heads) f32.  Needs output_heads.sequence_head.* in the state dict."""
        n = self.n_sequence_heads
        if not n:
            raise RuntimeError("this engine was built without output_heads.sequence_head.* weights")
        out = torch.empty(B, L, n, dtype=torch.float32, device=self.device)
        self._chk(self._lib.esmdiff_get_sequence_logits(self._h, _ptr(out), n, B, L, _stream()))
        return out

    def ddpm_step(self, x: torch.Tensor, logits: torch.Tensor, mc_t: float, mc_s: float, *, final: bool = False,
                  u: Optional[torch.Tensor] = None, seed: Optional[int] = None, sample_offset: int = 0,
                  step: int = 0) -> torch.Tensor:
        """In-place update of x (B,L) int64 from RAW logits (B,L,>=4101) float32 (last dim may be a strided
        view of a padded buffer).  Noise: explicit `u` (B,L,4101) or Philox(seed, sample_offset, step)."""
        B, L = x.shape
        assert x.dtype == torch.int64 and x.is_cuda and x.is_contiguous()
        assert logits.dtype == torch.float32 and logits.is_cuda and logits.stride(-1) == 1
        ld = logits.stride(1)
        assert logits.stride(0) == ld * L and ld >= STRUCTURE_VOCAB
        rng = None
        if u is not None:
            u = u.to(device=self.device, dtype=torch.float32).contiguous()
            assert u.shape == (B, L, STRUCTURE_VOCAB)
        elif not final:
            if seed is None:
                raise ValueError("ddpm_step needs explicit uniforms `u` or a Philox `seed`")
        if seed is not None:
            rng = N.Rng(int(seed), int(sample_offset))
        self._chk(self._lib.esmdiff_ddpm_step(self._h, _ptr(x), _ptr(logits), ld, float(mc_t), float(mc_s),
                                              int(final), _ptr(u), ctypes.byref(rng) if rng else None, int(step),
                                              B, L, _stream()))
        return x

    def ddpm_step_margin(self, x: torch.Tensor, logits: torch.Tensor, mc_t: float, mc_s: float, *, final: bool,
                         seed: int, sample_offset: int, step: int, margin: float, flags: torch.Tensor) -> torch.Tensor:
        """ddpm_step (Philox noise) that also sets flags[b] = 1 (int32 [B], zeroed by the caller) when some masked row of
        sample b was decided by less than `margin` (esmdiff_ddpm_step_margin; used by certified.py)."""
        B, L = x.shape
        assert x.dtype == torch.int64 and x.is_cuda and x.is_contiguous()
        assert logits.dtype == torch.float32 and logits.is_cuda and logits.stride(-1) == 1
        ld = logits.stride(1)
        assert logits.stride(0) == ld * L and ld >= STRUCTURE_VOCAB
        assert flags.dtype == torch.int32 and flags.is_cuda and flags.numel() == B and flags.is_contiguous()
        rng = N.Rng(int(seed), int(sample_offset))
        self._chk(self._lib.esmdiff_ddpm_step_margin(self._h, _ptr(x), _ptr(logits), ld, float(mc_t), float(mc_s), int(final),
                                                     ctypes.byref(rng), int(step), B, L, float(margin), _ptr(flags),
                                                     _stream()))
        return x

    @staticmethod
    def sample_step_params_host(sample_index, mc_t, mc_s, step, final):
        """Host sequences (one entry per sample) -> the esmdiff_sample_step records as a uint8 numpy array (n, 24)."""
        import numpy as np
        n = len(sample_index)
        rec = np.zeros(n, dtype=N.SAMPLE_STEP_DTYPE)
        rec["sample_index"], rec["step"], rec["final"] = sample_index, step, final
        rec["move_chance_t"], rec["move_chance_s"] = mc_t, mc_s
        return rec.view(np.uint8).reshape(n, -1)

    def sample_step_params(self, sample_index, mc_t, mc_s, step, final) -> torch.Tensor:
        """... as the device array ddpm_step_rows takes."""
        return torch.from_numpy(self.sample_step_params_host(sample_index, mc_t, mc_s, step, final)).to(self.device)

    def ddpm_step_rows(self, x: torch.Tensor, logits: torch.Tensor, params: torch.Tensor, *, seed: int,
                       eps: Optional[float] = None, flags: Optional[torch.Tensor] = None,
                       gaps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ddpm_step with one parameter set per sample (esmdiff_ddpm_step_rows): params from sample_step_params.  With `eps`
        (a bound on the logit error, certified.py) flags[b] (int32, zeroed by the caller) is set where a masked row of sample b
        was decided by less than exp(2 eps) (final passes: 2 eps in log p), and gaps[b] (f32, +inf on entry) gets the sample's
        smallest gap in log units."""
        import math
        B, L = x.shape
        assert x.dtype == torch.int64 and x.is_cuda and x.is_contiguous()
        assert logits.dtype == torch.float32 and logits.is_cuda and logits.stride(-1) == 1
        ld = logits.stride(1)
        assert logits.stride(0) == ld * L and ld >= STRUCTURE_VOCAB
        assert params.dtype == torch.uint8 and params.is_cuda and params.shape == (B, 24) and params.is_contiguous()
        for t_, dt in ((flags, torch.int32), (gaps, torch.float32)):
            assert t_ is None or (t_.dtype == dt and t_.is_cuda and t_.numel() == B and t_.is_contiguous())
        if (flags is not None or gaps is not None) and eps is None:
            raise ValueError("flags / gaps need eps")
        e2 = 0.0 if eps is None else 2.0 * float(eps)
        self._chk(self._lib.esmdiff_ddpm_step_rows(self._h, _ptr(x), _ptr(logits), ld, _ptr(params), int(seed), B, L,
                                                   math.exp(e2), e2, _ptr(flags), _ptr(gaps), _stream()))
        return x

    def logit_error_stats(self, a: torch.Tensor, b: torch.Tensor, x: torch.Tensor, all_columns: bool = False) -> torch.Tensor:
        """(n, L, 8) f32 = per token row {max |e|, sum e^2, max |d|, sum d^2, max e - min e, H(a) - H(b), H(b), 0}, e = a - b over the
        columns a decision reads — all but the MASK column (the ddpm draw), or every column with all_columns (the gibbs step) — of
        the rows of x (n, L) that are MASK; d = the error of neighbouring logit differences, the range bounds the error of ANY
        pair's difference, H = the row's softmax entropy; zeros elsewhere (esmdiff_logit_error_stats)."""
        n, L = x.shape
        for t_ in (a, b):
            assert t_.dtype == torch.float32 and t_.is_cuda and t_.stride(-1) == 1 and t_.shape[:2] == (n, L)
            assert t_.stride(0) == t_.stride(1) * L
        assert x.dtype == torch.int64 and x.is_cuda and x.is_contiguous()
        out = torch.empty(n, L, 8, dtype=torch.float32, device=self.device)
        N.check(self._lib.esmdiff_logit_error_stats(_ptr(a), a.stride(1), _ptr(b), b.stride(1), _ptr(x), n * L,
                                                    self.cfg.n_structure_heads, int(bool(all_columns)), _ptr(out), _stream()))
        return out

    def ddpm_sample(self, sequence_tokens: torch.Tensor, schedule: DDPMSchedule, *, seed: int,
                    sample_offset: int = 0, input_prior: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Whole ancestral sampling loop on the device with Philox noise (esmdiff_ddpm_sample)."""
        B, L = sequence_tokens.shape
        seq = self._tok(sequence_tokens, B, L)
        if input_prior is None:
            x = torch.full((B, L), STRUCTURE_MASK_TOKEN, dtype=torch.int64, device=self.device)
        else:
            if tuple(input_prior.shape) != (B, L):
                raise ValueError(f"Invalid input_prior shape: {tuple(input_prior.shape)} v.s. (seq) {(B, L)}")
            x = input_prior.to(device=self.device, dtype=torch.int64).contiguous().clone()
        self._check_ids(seq, x)
        T = schedule.num_steps
        f32 = lambda t: t.detach().to("cpu", torch.float32).contiguous()
        mc_t, mc_s = f32(schedule.mc_t[:T]), f32(schedule.mc_s[:T])
        tf = self.conditioning_rows(schedule.t_freq)
        tf = None if tf is None else f32(tf)
        rng = N.Rng(int(seed), int(sample_offset))
        as_p = lambda t: t.numpy().ctypes.data_as(N.c_f32p)
        self._chk(self._lib.esmdiff_ddpm_sample(self._h, _ptr(seq), _ptr(x), B, L, T, as_p(mc_t), as_p(mc_s),
                                                None if tf is None else as_p(tf),
                                                ctypes.byref(rng), _stream()))
        return x

    def gibbs_step(self, x: torch.Tensor, sequence_tokens: torch.Tensor, logits: torch.Tensor, temperature: float,
                   top_p: float, n_unmask: torch.Tensor, *, u: Optional[torch.Tensor] = None,
                   seed: Optional[int] = None, sample_offset: int = 0, step: int = 0) -> torch.Tensor:
        """One entropy-ordered unmasking step in place (esmdiff_gibbs_step).  n_unmask: (B,) int32."""
        B, L = x.shape
        assert x.dtype == torch.int64 and x.is_cuda and x.is_contiguous()
        assert logits.dtype == torch.float32 and logits.is_cuda and logits.stride(-1) == 1
        ld = logits.stride(1)
        seq = self._tok(sequence_tokens, B, L)
        nu = n_unmask.to(device=self.device, dtype=torch.int32).contiguous()
        rng = None
        if u is not None:
            u = u.to(device=self.device, dtype=torch.float32).contiguous()
            assert u.shape == (B, L, 4096)
        elif seed is None:
            raise ValueError("gibbs_step needs explicit uniforms `u` or a Philox `seed`")
        if seed is not None:
            rng = N.Rng(int(seed), int(sample_offset))
        self._chk(self._lib.esmdiff_gibbs_step(self._h, _ptr(x), _ptr(seq), _ptr(logits), ld, float(temperature),
                                               float(top_p), _ptr(nu), _ptr(u), ctypes.byref(rng) if rng else None,
                                               int(step), B, L, _stream()))
        return x

    @staticmethod
    def gibbs_step_params_host(sample_index, step, n_unmask):
        """Host sequences (one entry per prompt) -> the esmdiff_gibbs_sample_step records as a uint8 numpy array (n, 16)."""
        import numpy as np
        n = len(sample_index)
        rec = np.zeros(n, dtype=N.GIBBS_STEP_DTYPE)
        rec["sample_index"], rec["step"], rec["n_unmask"] = sample_index, step, n_unmask
        return rec.view(np.uint8).reshape(n, -1)

    def gibbs_step_rows(self, x: torch.Tensor, sequence_tokens: torch.Tensor, logits: torch.Tensor, temperature: float,
                        top_p: float, params: torch.Tensor, *, seed: int, pair_bound: Optional[float] = None,
                        entropy_bound: Optional[float] = None, flags: Optional[torch.Tensor] = None,
                        gaps: Optional[torch.Tensor] = None) -> torch.Tensor:
        """gibbs_step (Philox noise) with one parameter set per prompt (esmdiff_gibbs_step_rows): params from
        gibbs_step_params_host, uploaded.  With pair_bound R and entropy_bound E (bounds on the error of a row's logit differences
        and of its entropy, certified.py) flags[b] (int32, zeroed by the caller) gets bit 0 / 1 / 2 where the race / the nucleus
        membership / the entropy order of the rows prompt b unmasks could come out differently for logits within the bounds, and
        gaps[b] (f32 (B, 2), +inf on entry) the smallest race gap (logit units) and the selected-to-unselected entropy distance."""
        B, L = x.shape
        assert x.dtype == torch.int64 and x.is_cuda and x.is_contiguous()
        assert logits.dtype == torch.float32 and logits.is_cuda and logits.stride(-1) == 1
        ld = logits.stride(1)
        assert logits.stride(0) == ld * L
        seq = self._tok(sequence_tokens, B, L)
        assert params.dtype == torch.uint8 and params.is_cuda and params.shape == (B, 16) and params.is_contiguous()
        assert flags is None or (flags.dtype == torch.int32 and flags.is_cuda and flags.numel() == B and flags.is_contiguous())
        assert gaps is None or (gaps.dtype == torch.float32 and gaps.is_cuda and gaps.shape == (B, 2) and gaps.is_contiguous())
        if (flags is not None or gaps is not None) and pair_bound is None:
            raise ValueError("flags / gaps need pair_bound and entropy_bound")
        R = -1.0 if pair_bound is None else float(pair_bound)
        E = 0.0 if entropy_bound is None else float(entropy_bound)
        self._chk(self._lib.esmdiff_gibbs_step_rows(self._h, _ptr(x), _ptr(seq), _ptr(logits), ld, float(temperature), float(top_p),
                                                    _ptr(params), int(seed), B, L, R, E, _ptr(flags), _ptr(gaps), _stream()))
        return x

    def gibbs_sample(self, sequence_tokens: torch.Tensor, x0: torch.Tensor, n_unmask_table: torch.Tensor,
                     temperature: float, top_p: float, *, seed: int, sample_offset: int = 0) -> torch.Tensor:
        """Whole iterative-unmasking loop on the device (esmdiff_gibbs_sample).  n_unmask_table: (T,B) int32."""
        B, L = sequence_tokens.shape
        seq = self._tok(sequence_tokens, B, L)
        x = x0.to(device=self.device, dtype=torch.int64).contiguous().clone()
        self._check_ids(seq, x)
        tab = n_unmask_table.detach().to("cpu", torch.int32).contiguous()
        T = tab.shape[0]
        assert tab.shape == (T, B)
        rng = N.Rng(int(seed), int(sample_offset))
        self._chk(self._lib.esmdiff_gibbs_sample(self._h, _ptr(seq), _ptr(x), B, L, T, float(temperature), float(top_p),
                                                 tab.numpy().ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                 ctypes.byref(rng), _stream()))
        return x

    def set_frames(self, rot: Optional[torch.Tensor], trans: Optional[torch.Tensor] = None,
                   has_frame: Optional[torch.Tensor] = None) -> None:
        """Coordinate conditioning for the following forwards (esmdiff_set_frames): rot (B,L,3,3), trans (B,L,3),
        has_frame (B,L) bool — see esmdiff_amd.geometry.build_affine3d_from_coordinates.  None clears them."""
        if rot is None:
            self._chk(self._lib.esmdiff_set_frames(self._h, None, None, None, 0, 0, _stream()))
            self._frames = None
            return
        B, L = rot.shape[:2]
        r = rot.to(device=self.device, dtype=torch.float32).contiguous()
        t = trans.to(device=self.device, dtype=torch.float32).contiguous()
        m = has_frame.to(device=self.device, dtype=torch.uint8).contiguous()
        assert r.shape == (B, L, 3, 3) and t.shape == (B, L, 3) and m.shape == (B, L)
        self._chk(self._lib.esmdiff_set_frames(self._h, _ptr(r), _ptr(t), _ptr(m), B, L, _stream()))
        self._frames = (r, t, m)   # keep the staging tensors alive until the async copies have run

    def gemm(self, A: torch.Tensor, W: torch.Tensor, epilogue: int, *, bias: Optional[torch.Tensor] = None,
             alpha: float = 1.0) -> torch.Tensor:
        """bf16-output GEMM exactly as the forward issues it (esmdiff_gemm_bf16_ws: split-K workspace attached)."""
        M, K = A.shape
        Nn = W.shape[0]
        assert A.dtype == W.dtype == torch.bfloat16 and A.is_contiguous() and W.is_contiguous() and W.shape[1] == K
        out = torch.empty(M, Nn // 2 if epilogue == N.EPI_SWIGLU_BF16 else Nn, dtype=torch.bfloat16, device=A.device)
        self._chk(self._lib.esmdiff_gemm_bf16_ws(self._h, _ptr(A), _ptr(W), _ptr(out), _ptr(bias), M, Nn, K, out.stride(0),
                                                 Nn, float(alpha), epilogue, _stream()))
        return out

    def set_gibbs_options(self, strategy: str = "entropy", invalid_ids=()) -> None:
        """GenerationConfig.strategy ("entropy" | "random") and .invalid_ids for the following gibbs steps
        (esmdiff_set_gibbs_options); the defaults restore esm's default behaviour."""
        if strategy not in ("entropy", "random"):
            raise ValueError(f"strategy must be 'entropy' or 'random', got {strategy!r}")
        ids = [int(v) for v in invalid_ids]
        arr = (ctypes.c_int32 * max(1, len(ids)))(*ids)
        self._chk(self._lib.esmdiff_set_gibbs_options(self._h, 1 if strategy == "random" else 0, arr if ids else None, len(ids)))

    def set_streams(self, n_streams: int, min_tokens: Optional[int] = None) -> None:
        """Sub-batch launch queues of the 16-bit forward (esmdiff_set_option: 1 .. 4, default 2) and the token count from which a
        batch is cut into sub-batches (default 2200).  No result bit depends on either: the dispatch path of a (B, L) forward is the
        default options' choice, a setting that would move the sub-batches onto the other path is reduced (esmdiff_set_option)."""
        self._chk(self._lib.esmdiff_set_option(self._h, N.OPT_STREAMS, int(n_streams)))
        if min_tokens is not None:
            self._chk(self._lib.esmdiff_set_option(self._h, N.OPT_DUAL_MIN_TOKENS, int(min_tokens)))

    def describe_plan(self, B: int, L: int) -> str:
        """How a (B, L) forward will be dispatched on this engine, as text (esmdiff_describe_plan)."""
        buf = ctypes.create_string_buffer(2048)
        n = self._lib.esmdiff_describe_plan(self._h, int(B), int(L), buf, 2048)
        if n < 0:
            self._chk(n)
        return buf.value.decode()

    def set_step0_sharing(self, on: bool) -> None:
        """Exact step-0 sharing (esmdiff_set_step0_sharing): when every sample of a ddpm_sample / gibbs_sample call starts
        from identical tokens (checked on the device), the first forward runs on a sub-batch and serves all samples; ids are
        bit-identical to the unshared run.  Off by default."""
        self._chk(self._lib.esmdiff_set_step0_sharing(self._h, int(bool(on))))

    def shared_forward_batch(self, B: int, L: int) -> int:
        """Smallest n <= B whose (n, L) forward takes the dispatch path of a (B, L) forward — bit-identical logits per sample
        (esmdiff_shared_forward_batch); B when there is none."""
        n = self._lib.esmdiff_shared_forward_batch(self._h, int(B), int(L))
        if n < 0:
            self._chk(n)
        return int(n)

    def set_small_batch_splitk(self, on: bool) -> None:
        """F32_SPLIT engines: K-sliced residual linears for forwards of <= 4096 rows (esmdiff_set_small_batch_splitk) — faster
        small batches, at the price of the bit-for-bit batch independence across that row count.  Off by default."""
        self._chk(self._lib.esmdiff_set_small_batch_splitk(self._h, int(bool(on))))

    def set_final_skip(self, on: bool) -> None:
        """Exact skip of the noise-removal forward (esmdiff_set_final_skip): after the last update only the samples that still
        hold a MASK run forward T + 1 (none, almost always); ids are bit-identical to the full run.  Off by default."""
        self._chk(self._lib.esmdiff_set_final_skip(self._h, int(bool(on))))

    def counters(self, reset: bool = False) -> Dict[str, int]:
        """Network forwards issued and token rows pushed through them since create / the last reset (executed work)."""
        f, r = ctypes.c_int64(0), ctypes.c_int64(0)
        self._chk(self._lib.esmdiff_get_counters(self._h, ctypes.byref(f), ctypes.byref(r), int(reset)))
        return {"forwards": int(f.value), "token_rows": int(r.value)}

    # ---- per-kernel entry points (parity tests / roofline bench) ---------------------------------
    def set_profiling(self, mode):
        """0/False off; 1/True HIP events around every launch; 2 only around the dominant kernel (FFN-up GEMM)."""
        self._chk(self._lib.esmdiff_set_profiling(self._h, int(mode)))

    def get_profile(self) -> Dict[str, Dict[str, float]]:
        ms = (ctypes.c_float * 16)()
        n = (ctypes.c_int32 * 16)()
        self._chk(self._lib.esmdiff_get_profile(self._h, ms, n))
        out = {s: {"ms": float(ms[i]), "launches": int(n[i])} for i, s in enumerate(N.SECTIONS)}
        # profiling mode 2: union of the FFN-up launches' busy intervals over all streams (slot 15 of the C ABI arrays)
        out["gemm_ffn_up_union"] = {"ms": float(ms[15]), "launches": int(n[15])}
        return out

    def attention(self, qkv: torch.Tensor, q_ln_w: torch.Tensor, k_ln_w: torch.Tensor, B: int, L: int) -> torch.Tensor:
        D = self.cfg.d_model
        assert qkv.dtype == torch.bfloat16 and qkv.shape == (B * L, 3 * D) and qkv.is_contiguous()
        ctx = torch.empty(B * L, D, dtype=torch.bfloat16, device=self.device)
        self._chk(self._lib.esmdiff_attention_bf16(self._h, _ptr(qkv), _ptr(q_ln_w.float().contiguous()),
                                                   _ptr(k_ln_w.float().contiguous()), _ptr(ctx), B, L, _stream()))
        return ctx


class StructureDecoder:
    """Structure tokens -> backbone coordinates on the device (esmdiff_decoder_create / esmdiff_decoder_decode): what the
    reference gets from `esm3.decode(ESMProteinTensor(structure=...))`, /root/reference/slm/sample_esmdiff.py:40-61."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], max_batch: int, max_len: int, device: int = 0,
                 precision: str = "f32"):
        """precision defaults to "f32": the reference decodes in the model's float32 and north_star's bar for this output
        is a backbone within 1e-4 A; decoding is 3e13 FLOP per 100 samples and sits outside the timed sampling metric.
        "bf16" is the r01 / r02 MFMA path (0.05 A mean off the f32 result, ~10x faster)."""
        _require_gpu()
        if precision not in N.PRECISION:
            raise ValueError(f"precision must be one of {sorted(N.PRECISION)}, got {precision!r}")
        self.precision = precision
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self._lib = N.lib()
        self._h = ctypes.c_void_p(0)
        c = N.Config(cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.ffn_hidden, 23, 1, max_batch, max_len, 1.0, 0,
                     N.PRECISION[precision], 0)
        keep, table = [], (N.Weight * len(state_dict))()
        with torch.cuda.device(self.device):
            for i, (name, t) in enumerate(state_dict.items()):
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                d = t.detach().to(self.device).contiguous()
                keep.append(d)
                shape = (ctypes.c_int64 * 4)(*(list(d.shape) + [0] * (4 - d.dim())))
                table[i] = N.Weight(name.encode(), d.data_ptr(), N.DT_F32 if d.dtype == torch.float32 else N.DT_BF16,
                                    d.dim(), shape)
            torch.cuda.synchronize()
            code = self._lib.esmdiff_decoder_create(ctypes.byref(c), table, len(state_dict), device, ctypes.byref(self._h))
        if code != 0:
            raise RuntimeError(f"esmdiff_decoder_create failed ({code}): {self._lib.esmdiff_last_error(None).decode()}")
        del keep
        self.max_batch, self.max_len = max_batch, max_len
        self.has_plddt = any(k == "plddt_head.3.weight" for k in state_dict)
        self.has_ptm = any(k == "pairwise_classification_head.linear2.weight" for k in state_dict)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.esmdiff_engine_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, structure_tokens: torch.Tensor, return_plddt: bool = False, return_ptm: bool = False,
               return_pae: bool = False):
        """structure_tokens (B, L) int64 including BOS / EOS -> backbone coordinates (B, L - 2, 3, 3) float32 (N, CA, C);
        with return_plddt also the per-residue pLDDT (B, L - 2) in [0, 1] (None when the weights carry no plddt_head);
        with return_ptm / return_pae also pTM (B,) and the predicted aligned error (B, L, L, BOS / EOS rows kept, as esm
        returns it) — None when the weights carry no pairwise_classification_head.  Return order: coords[, plddt][, ptm][, pae]."""
        B, L = structure_tokens.shape
        tok = structure_tokens.to(device=self.device, dtype=torch.int64).contiguous()
        if int(tok.min()) < 0 or int(tok.max()) >= STRUCTURE_VOCAB:
            raise ValueError(f"structure token id out of range 0..{STRUCTURE_VOCAB - 1}")
        out = torch.empty(B, L, 3, 3, dtype=torch.float32, device=self.device)
        pl = torch.empty(B, L, dtype=torch.float32, device=self.device) if (return_plddt and self.has_plddt) else None
        want_pair = (return_ptm or return_pae) and self.has_ptm
        ptm = torch.empty(B, dtype=torch.float32, device=self.device) if want_pair else None
        pae = torch.empty(B, L, L, dtype=torch.float32, device=self.device) if (return_pae and self.has_ptm) else None
        N.check(self._lib.esmdiff_decoder_decode(self._h, _ptr(tok), _ptr(out), _ptr(pl), _ptr(ptm), _ptr(pae), B, L,
                                                 float(self.cfg.trans_scale), _stream()), self._h)
        res = [out[:, 1:-1]]
        if return_plddt:
            res.append(None if pl is None else pl[:, 1:-1])
        if return_ptm:
            res.append(ptm)
        if return_pae:
            res.append(pae)
        return res[0] if len(res) == 1 else tuple(res)


class StructureEncoder:
    """Backbone coordinates -> structure tokens on the device (esmdiff_encoder_create / esmdiff_encoder_encode): what the
    reference gets from `model.encode(ESMProtein(coordinates=...))`, /root/reference/slm/models/utils.py:136-137."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], device: int = 0, precision: str = "f32"):
        """precision defaults to "f32" (csrc/strict.hip): the encoder runs once per input structure on B*L*16 neighbourhood
        rows, and its output is a nearest-code search where bf16 GEMM noise flips 1-2 % of the codes at near-ties."""
        _require_gpu()
        if precision not in N.PRECISION:
            raise ValueError(f"precision must be one of {sorted(N.PRECISION)}, got {precision!r}")
        self.precision = precision
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self._lib = N.lib()
        self._h = ctypes.c_void_p(0)
        keep, table = [], (N.Weight * len(state_dict))()
        with torch.cuda.device(self.device):
            for i, (name, t) in enumerate(state_dict.items()):
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                d = t.detach().to(self.device).contiguous()
                keep.append(d)
                shape = (ctypes.c_int64 * 4)(*(list(d.shape) + [0] * (4 - d.dim())))
                table[i] = N.Weight(name.encode(), d.data_ptr(), N.DT_F32 if d.dtype == torch.float32 else N.DT_BF16,
                                    d.dim(), shape)
            torch.cuda.synchronize()
            code = self._lib.esmdiff_encoder_create(cfg.d_model, cfg.v_heads, cfg.n_layers, cfg.ffn_hidden, cfg.d_out,
                                                    cfg.n_codes, cfg.knn, cfg.relpos_bins, N.PRECISION[precision], table,
                                                    len(state_dict), device,
                                                    ctypes.byref(self._h))
        if code != 0:
            raise RuntimeError(f"esmdiff_encoder_create failed ({code}): {self._lib.esmdiff_encoder_last_error(None).decode()}")
        del keep

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.esmdiff_encoder_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, coordinates: torch.Tensor) -> torch.Tensor:
        """coordinates (B, L, >=3, 3) with N, CA, C first (NaN / Inf where unknown) -> structure tokens (B, L) int64 on the
        device; MASK (4096) where a residue has no coordinates.  BOS / EOS are the caller's to add."""
        from .geometry import build_affine3d_from_coordinates
        rot, trans, has = build_affine3d_from_coordinates(coordinates)
        B, L = has.shape
        ca = torch.where(has[..., None], coordinates[..., 1, :].to(torch.float32), torch.zeros(B, L, 3))
        dev = lambda t, dt: t.to(device=self.device, dtype=dt).contiguous()  # noqa: E731
        ca, rot, trans, hm = dev(ca, torch.float32), dev(rot, torch.float32), dev(trans, torch.float32), dev(has, torch.uint8)
        tok = torch.empty(B, L, dtype=torch.int64, device=self.device)
        code = self._lib.esmdiff_encoder_encode(self._h, _ptr(ca), _ptr(rot), _ptr(trans), _ptr(hm), _ptr(tok), B, L, _stream())
        if code != 0:
            raise RuntimeError(f"esmdiff_encoder_encode failed ({code}): {self._lib.esmdiff_encoder_last_error(self._h).decode()}")
        return tok


def gemm_bf16(A: torch.Tensor, W: torch.Tensor, epilogue: int, *, out: Optional[torch.Tensor] = None,
              bias: Optional[torch.Tensor] = None, alpha: float = 1.0, n_valid: Optional[int] = None) -> torch.Tensor:
    """out = epilogue(A[M,K] @ W[N,K]^T) through esmdiff_gemm_bf16 (N % 128 == 0, K % 64 == 0)."""
    _require_gpu()
    M, K = A.shape
    Nn = W.shape[0]
    assert A.dtype == W.dtype == torch.bfloat16 and A.is_contiguous() and W.is_contiguous() and W.shape[1] == K
    if out is None:
        if epilogue in (N.EPI_BF16, N.EPI_BIAS_GELU_BF16):
            out = torch.empty(M, Nn, dtype=torch.bfloat16, device=A.device)
        elif epilogue == N.EPI_SWIGLU_BF16:
            out = torch.empty(M, Nn // 2, dtype=torch.bfloat16, device=A.device)
        else:
            raise ValueError("f32 epilogues need an explicit `out`")
    ldc = out.stride(0)
    N.check(N.lib().esmdiff_gemm_bf16(_ptr(A), _ptr(W), _ptr(out), _ptr(bias), M, Nn, K, ldc,
                                      Nn if n_valid is None else n_valid, float(alpha), epilogue, _stream()))
    return out


def gemm_f16(A: torch.Tensor, W: torch.Tensor, epilogue: int, *, out: Optional[torch.Tensor] = None,
             bias: Optional[torch.Tensor] = None, alpha: float = 1.0, n_valid: Optional[int] = None) -> torch.Tensor:
    """gemm_bf16's contract on the f16 build of the kernels (esmdiff_gemm_f16): A, W and 16-bit outputs are torch.float16."""
    _require_gpu()
    M, K = A.shape
    Nn = W.shape[0]
    assert A.dtype == W.dtype == torch.float16 and A.is_contiguous() and W.is_contiguous() and W.shape[1] == K
    if out is None:
        if epilogue in (N.EPI_BF16, N.EPI_BIAS_GELU_BF16):
            out = torch.empty(M, Nn, dtype=torch.float16, device=A.device)
        elif epilogue == N.EPI_SWIGLU_BF16:
            out = torch.empty(M, Nn // 2, dtype=torch.float16, device=A.device)
        else:
            raise ValueError("f32 epilogues need an explicit `out`")
    N.check(N.lib().esmdiff_gemm_f16(_ptr(A), _ptr(W), _ptr(out), _ptr(bias), M, Nn, K, out.stride(0),
                                     Nn if n_valid is None else n_valid, float(alpha), epilogue, _stream()))
    return out


def gemm_f32(A: torch.Tensor, W: torch.Tensor, epilogue: int = 0, *, out: Optional[torch.Tensor] = None,
             bias: Optional[torch.Tensor] = None, div: float = 1.0) -> torch.Tensor:
    """The strict path's linear (esmdiff_gemm_f32): out f32 [M,N] = epi(A f32 [M,K] @ W f32 [N,K]^T); K % 32 == 0.
    epilogue N.F32EPI_RESID_DIV updates `out` in place: out + acc / div."""
    _require_gpu()
    M, K = A.shape
    Nn = W.shape[0]
    assert A.dtype == W.dtype == torch.float32 and A.stride(1) == 1 and W.is_contiguous() and W.shape[1] == K
    if out is None:
        if epilogue == N.F32EPI_RESID_DIV:
            raise ValueError("the residual epilogue updates `out` in place: pass it")
        out = torch.empty(M, Nn, dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32 and out.stride(1) == 1
    N.check(N.lib().esmdiff_gemm_f32(_ptr(A), A.stride(0), _ptr(W), _ptr(out), _ptr(bias), M, Nn, K, out.stride(0), Nn,
                                     float(div), epilogue, _stream()))
    return out


def split_rows(A: torch.Tensor):
    """f32 [M,K] -> (a3 f16 [M,3K] = [hi | lo | hi] of the row scaled by a power of two, rs f32 [M] = 1 / scale): the
    activation operand of the F32_SPLIT linears (esmdiff_split_rows)."""
    _require_gpu()
    M, K = A.shape
    assert A.dtype == torch.float32 and A.stride(1) == 1
    a2 = torch.empty(M, 3 * K, dtype=torch.float16, device=A.device)
    rs = torch.empty(M, dtype=torch.float32, device=A.device)
    N.check(N.lib().esmdiff_split_rows(_ptr(A), A.stride(0), _ptr(a2), _ptr(rs), M, K, _stream()))
    return a2, rs


def split_weight(W: torch.Tensor):
    """f32 [N,K] -> (w3 f16 [N_pad,3K] = [lo | hi | hi] of W * 2^k, rows padded with zeros to a multiple of 256, 2^-k)."""
    _require_gpu()
    Nn, K = W.shape
    assert W.dtype == torch.float32 and W.is_contiguous()
    n_pad = (Nn + 255) // 256 * 256
    w2 = torch.empty(n_pad, 3 * K, dtype=torch.float16, device=W.device)
    inv = ctypes.c_float(0)
    torch.cuda.synchronize()
    N.check(N.lib().esmdiff_split_weight(_ptr(W), _ptr(w2), Nn, n_pad, K, ctypes.byref(inv)))
    return w2, float(inv.value)


def gemm_split(a2: torch.Tensor, rs: Optional[torch.Tensor], w2: torch.Tensor, w_inv: float, n_out: int,
               epilogue: int = 0, *, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
               div: float = 1.0) -> torch.Tensor:
    """out f32 [M, n_out] = epi(A . W^T) from split operands on three f16 MFMA passes (esmdiff_gemm_split)."""
    _require_gpu()
    M, K3 = a2.shape
    n_pad = w2.shape[0]
    assert a2.dtype == w2.dtype == torch.float16 and w2.shape[1] == K3 and a2.is_contiguous() and w2.is_contiguous()
    if out is None:
        if epilogue == N.F32EPI_RESID_DIV:
            raise ValueError("the residual epilogue updates `out` in place: pass it")
        out = torch.empty(M, n_pad, dtype=torch.float32, device=a2.device)      # the kernel writes whole padded rows
    assert out.dtype == torch.float32 and out.stride(1) == 1 and out.stride(0) >= n_pad
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= n_pad
    N.check(N.lib().esmdiff_gemm_split(_ptr(a2), _ptr(rs), _ptr(w2), float(w_inv), _ptr(out), _ptr(bias), M, n_pad, K3 // 3,
                                       out.stride(0), float(div), epilogue, _stream()))
    return out[:, :n_out]


def gemm_bf16_timed(A, W, out, epilogue, iters=20, bias=None, alpha=1.0) -> float:
    """Average milliseconds per launch, HIP events on the launch stream (esmdiff_gemm_bf16_timed; -DED_DEBUG builds only)."""
    _require_gpu()
    if not hasattr(N.lib(), "esmdiff_gemm_bf16_timed"):
        raise RuntimeError("esmdiff_gemm_bf16_timed is a measurement aid of -DED_DEBUG builds: "
                           "ESMDIFF_EXTRA_CXXFLAGS=-DED_DEBUG python -m esmdiff_amd.build --force")
    M, K = A.shape
    ms = ctypes.c_float(0)
    N.check(N.lib().esmdiff_gemm_bf16_timed(_ptr(A), _ptr(W), _ptr(out), _ptr(bias), M, W.shape[0], K,
                                            out.stride(0), W.shape[0], float(alpha), epilogue, iters,
                                            ctypes.byref(ms), _stream()))
    return float(ms.value)


def layernorm_bf16(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    _require_gpu()
    M, D = x.shape
    y = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    N.check(N.lib().esmdiff_layernorm_bf16(_ptr(x.float().contiguous()), _ptr(w.float().contiguous()),
                                           _ptr(None if b is None else b.float().contiguous()), _ptr(y), M, D,
                                           _stream()))
    return y
