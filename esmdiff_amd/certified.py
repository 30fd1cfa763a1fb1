"""Certified sampling: the ids of the f32-grade chain at close to the reduced-precision engine's speed.

The state of the ancestral sampler (model.py:543-581) between two updates is a matrix of token ids, so a rounding error of
the network does not accumulate from update to update: as long as every draw of an update came out the same, the next
update starts from the identical input.  A draw is an arg-max of q_v / g_v (model.py:24-28) with q_v = exp(z_v - lse) *
(mc_t - mc_s), or mc_s for the mask column.  With logits known to +-eps, the ratio of two candidates moves by at most
exp(2 eps): between two tokens the logsumexp cancels and z_a - z_b moves by 2 eps; against the mask column z_a moves by
eps and lse (1-Lipschitz in the max norm) by eps.  So a winner that beats the runner-up by more than the factor
exp(2 eps) is the winner for ANY logits within eps — in particular for the f32 ones.  The sampler kernel reports,
per sample, whether all of its draws were that clear (esmdiff_ddpm_step_margin, csrc/sampler.hip).  The few samples with
a close call are run again through the f32-grade engine for that one update (same tokens in, same Philox keys) and take
its ids.

`eps` is a bound on the fast engine's logit error against f32, and it is an EMPIRICAL one: the result is "the f32 chain's
ids unless a logit was off by more than eps", which tests/test_gpu_strict.py checks at configs[1]'s full size (100
samples, 335 340 draws: all ids equal).  It is measured while sampling: every re-run yields the fast and the f32-grade
logits of the same input (258 x 4101 of them per sample), `max_logit_err_observed` is their largest difference over the
masked rows, and with eps=None (the default) the bound used for the NEXT update is `safety` (2.0) x the largest error
seen so far by this engine pair — the logit scale of a checkpoint, and with it the error, is not known in advance (random
init: logit std 0.6, f16 error up to 2.1e-3; with the f32-grade head 1.3e-3).  Before the first certified update two
samples are run on both engines to start the estimate.  stats["eps_violations"] counts re-run samples whose error
exceeded the eps their update was certified with.  No reference counterpart — the reference has one precision.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .engine import Engine
from .schedule import DDPMSchedule
from .constants import STRUCTURE_MASK_TOKEN


class CertifiedSampler:
    """fast: a reduced-precision Engine (f16 recommended: its logit error is 8x below bf16's, so 8x fewer close calls);
    exact: an f32-grade Engine of the same checkpoint (precision 'f32_split' or 'f32')."""

    def __init__(self, fast: Engine, exact: Engine, eps: Optional[float] = None, safety: float = 2.0, eps_floor: float = 1e-5,
                 fast_reruns: bool = False):
        if fast.device != exact.device:
            raise ValueError("both engines must live on the same GPU")
        if eps is not None and not eps > 0:
            raise ValueError("eps must be positive")
        if not safety >= 1.0:
            raise ValueError("safety must be >= 1")
        self.fast, self.exact = fast, exact
        # fast_reruns: the re-runs are small batches, and the F32_SPLIT engine can run their residual linears K-sliced
        # (esmdiff_set_small_batch_splitk: 2.8 -> 2.6 s at configs[1]).  That is another float32-grade evaluation — its logits
        # differ from the same engine's large-batch ones in the last bits (<= 1.1e-6) — so a draw tied to within that can come out
        # differently: over 30 seeds at configs[1] one id of one run differed from the engine's own chain (and none with the
        # option off, where the certified chain IS the exact engine's chain as long as eps holds).  Off by default.
        self.fast_reruns = bool(fast_reruns)
        self.eps = None if eps is None else float(eps)          # None: safety x the largest error observed so far
        self.safety, self.eps_floor = float(safety), float(eps_floor)
        self.err_seen = 0.0                                     # largest |fast - exact| logit over masked rows, all calls
        self.n_seen = 0                                         # sample-forwards that estimate rests on
        self.stats: dict = {}

    def _eps_now(self) -> float:
        return self.eps if self.eps is not None else max(self.eps_floor, self.safety * self.err_seen)

    def _observe(self, lg_fast: torch.Tensor, lg_exact: torch.Tensor, x_in: torch.Tensor) -> torch.Tensor:
        """Largest logit difference per sample over its masked rows; feeds the running estimate."""
        e = ((lg_fast - lg_exact).abs().amax(-1) * (x_in == STRUCTURE_MASK_TOKEN)).amax(-1)
        self.err_seen = max(self.err_seen, float(e.max()))
        self.n_seen += int(e.numel())
        return e

    def _rerun(self, i, idx, xs, sq, lg_fast, schedule, tf_exact, seed, sample_offset):
        """Update i of the samples `idx` on the f32-grade engine (current stream): xs (their tokens before the update) -> their
        certified tokens; also the largest fast-vs-exact logit difference per sample over its masked rows."""
        exact, T = self.exact, schedule.num_steps
        fin = i == T
        mc_t = 0.0 if fin else float(schedule.mc_t[i])
        mc_s = 0.0 if fin else float(schedule.mc_s[i])
        out, errs = xs.clone(), []
        for c0 in range(0, xs.shape[0], exact.max_batch):
            sl = slice(c0, c0 + exact.max_batch)
            lg2 = exact.forward_logits(xs[sl], sq[sl], None if tf_exact is None else tf_exact[i])
            errs.append(((lg_fast[sl] - lg2).abs().amax(-1) * (xs[sl] == STRUCTURE_MASK_TOKEN)).amax(-1))
            for j, b in enumerate(idx[sl].tolist()):
                exact.ddpm_step(out[c0 + j:c0 + j + 1], lg2[j:j + 1], mc_t, mc_s, final=fin, seed=seed,
                                sample_offset=sample_offset + b, step=i)
        return out, torch.cat(errs)

    def _account(self, e: torch.Tensor, eps_i: float, acc: dict) -> None:
        """Error observations of a re-run: the running estimate, this call's maximum, the eps violations."""
        m = float(e.max())
        self.err_seen = max(self.err_seen, m)
        self.n_seen += int(e.numel())
        acc["err"] = max(acc["err"], m)
        acc["viol"] += int((e > eps_i).sum())

    @torch.no_grad()
    def ddpm_sample(self, sequence_tokens: torch.Tensor, schedule: DDPMSchedule, *, seed: int, sample_offset: int = 0,
                    input_prior: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same arguments and result layout as Engine.ddpm_sample.  self.stats holds the re-run counts of the call."""
        fast, exact = self.fast, self.exact
        if getattr(exact, "precision", None) == "f32_split":      # (per call: two samplers may share one f32-grade engine)
            exact.set_small_batch_splitk(self.fast_reruns)
        B, L = sequence_tokens.shape
        dev = fast.device
        seq = sequence_tokens.to(device=dev, dtype=torch.int64).contiguous()
        if input_prior is None:
            x = torch.full((B, L), STRUCTURE_MASK_TOKEN, dtype=torch.int64, device=dev)
        else:
            if tuple(input_prior.shape) != (B, L):
                raise ValueError(f"Invalid input_prior shape: {tuple(input_prior.shape)} v.s. (seq) {(B, L)}")
            x = input_prior.to(device=dev, dtype=torch.int64).contiguous().clone()
        T = schedule.num_steps
        tf_fast = fast.conditioning_rows(schedule.t_freq)
        tf_exact = exact.conditioning_rows(schedule.t_freq)
        flags = torch.zeros(B, dtype=torch.int32, device=dev)
        logits = torch.empty(B, L, fast.ld_logits, dtype=torch.float32, device=dev)
        cap = exact.max_batch
        reruns = []
        skipped_final, probes_exact, probes_fast, eps_used = False, 0, 0, []
        acc = {"err": 0.0, "viol": 0}                               # largest logit error seen in this call, eps violations
        V = exact.cfg.n_structure_heads
        shared0 = input_prior is None and B > 1 and bool((seq == seq[:1]).all())
        for i in range(T + 1):
            fin = i == T
            if i == 0 and shared0 and not fin:
                # every sample enters the first update with the same inputs (all-mask prior, one protein): ONE f32-grade
                # forward serves them all, nothing to certify (the device loop's step-0 sharing, engine.hip)
                lg1 = exact.forward_logits(x[:1], seq[:1], None if tf_exact is None else tf_exact[0])
                if self.eps is None and self.n_seen == 0:          # start the error estimate on the same input
                    lgf = fast.forward_logits(x[:1], seq[:1], None if tf_fast is None else tf_fast[0])
                    acc["err"] = max(acc["err"], float(self._observe(lgf, lg1, x[:1]).max()))
                    probes_fast += 1
                logits[..., :V] = lg1
                exact.ddpm_step(x, logits[..., :V], float(schedule.mc_t[0]), float(schedule.mc_s[0]), seed=seed,
                                sample_offset=sample_offset, step=0)
                reruns.append(0)
                eps_used.append(None)
                continue
            if fin and not bool((x == STRUCTURE_MASK_TOKEN).any()):
                reruns.append(0)                                  # nothing left to denoise: the pass is the identity
                eps_used.append(None)
                skipped_final = True
                break
            mc_t = 0.0 if fin else float(schedule.mc_t[i])
            mc_s = 0.0 if fin else float(schedule.mc_s[i])
            prev = x.clone()
            lg = fast.forward_logits(x, seq, None if tf_fast is None else tf_fast[i], out=logits)
            if self.eps is None and self.n_seen < 3:              # thin estimate: two samples on both engines first
                n = min(2, B)
                lgp = exact.forward_logits(prev[:n], seq[:n], None if tf_exact is None else tf_exact[i])
                acc["err"] = max(acc["err"], float(self._observe(lg[:n], lgp, prev[:n]).max()))
                probes_exact += n
            eps_i = self._eps_now()
            eps_used.append(eps_i)
            ratio, diff = math.exp(2.0 * eps_i), 2.0 * eps_i
            flags.zero_()
            fast.ddpm_step_margin(x, lg, mc_t, mc_s, final=fin, seed=seed, sample_offset=sample_offset, step=i,
                                  margin=diff if fin else ratio, flags=flags)
            sus = torch.nonzero(flags).flatten()                  # (device -> host sync, B int32)
            reruns.append(int(sus.numel()))
            if int(sus.numel()) == 0:
                continue
            xs, sq, lgf = prev[sus], seq[sus], lg[sus]                       # (index gathers: fresh contiguous tensors)
            xr, e = self._rerun(i, sus, xs, sq, lgf, schedule, tf_exact, seed, sample_offset)
            self._account(e, eps_i, acc)
            x[sus] = xr
        err_max, violations = acc["err"], acc["viol"]
        used = [e_ for e_ in eps_used if e_ is not None]
        self.stats = {"samples": B, "updates": len(reruns), "eps": self.eps if self.eps is not None else "auto",
                      "safety": self.safety, "eps_min_used": min(used) if used else None, "eps_max_used": max(used) if used else None,
                      "rerun_per_update": reruns, "max_logit_err_observed": err_max, "max_logit_err_all_calls": self.err_seen,
                      "eps_violations": violations, "first_update_shared": shared0, "fast_reruns": self.fast_reruns,
                      "sample_forwards_exact": int(sum(reruns)) + int(shared0) + probes_exact,
                      "sample_forwards_fast": B * (len(reruns) - int(shared0) - int(skipped_final)) + probes_fast}
        return x
