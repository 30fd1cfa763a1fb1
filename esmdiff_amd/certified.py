"""Certified sampling: the ids of the f32-grade chain at close to the reduced-precision engine's speed, for both sampling
modes of the CLI (ddpm: /root/reference/slm/models/model.py:543-607; gibbs: /root/reference/slm/sample_esmdiff.py:66-130, the
reference's default).

The state of either sampler between two updates is a matrix of token ids, so a rounding error of the network does not
accumulate from update to update: as long as every decision of an update came out the same, the next update starts from the
identical input.  What an update decides, and what can move it:

  ddpm    a draw is an arg-max of q_v / g_v (model.py:24-28) with q_v = exp(z_v - lse) * (mc_t - mc_s), or mc_s for the mask
          column.  Between two tokens the logsumexp cancels, so the log of the ratio of two candidates moves by the error of the
          logit DIFFERENCE z_a - z_b; against the mask column it moves by the error of z_a - lse, and lse is a convex combination
          of the logits.  Both are at most the RANGE of the row's logit error, max e - min e.  If that is at most P, a winner that
          beats the runner-up by more than the factor exp(P) is the winner for the f32-grade logits as well
          (esmdiff_ddpm_step_rows, csrc/sampler.hip).
  gibbs   three decisions per step (esm's iterative_sampling_raw): the nucleus cut (a token's membership
          moves only if the cumulative mass at it is within the factor exp(+-P) of top_p, or a token within P of it changes
          sides), the temperature race among the kept tokens (as above, with P / temperature), and which k positions have the
          lowest entropy (moves only if the k-th and the (k+1)-th entropy are within 2 E, E a bound on the entropy error).  Only
          the rows a prompt unmasks in this step count.  esmdiff_gibbs_step_rows (csrc/gibbs.hip) reports all three per prompt.

What happens to a sample with a close call (speculative, batched, audited).  The samples of a batch are independent, so nothing
forces a flagged sample to be settled in the update it was flagged in:

  fast lane   every sample carries its own update counter; one reduced-precision forward per iteration advances all unfinished
              samples (ddpm: one sigma per sample, esmdiff_forward_logits_sigmas), flagged or not — a flagged sample CONTINUES on
              its fast ids, speculatively.  Nothing in the lane waits for the host: flags come back through pinned memory one
              iteration late, while the GPU is already inside the next forward.
  slow lane   a flagged sample-update (tokens before, fast tokens after, fast logits) is queued; when `verify_batch` of them
              have gathered they are evaluated in ONE f32-grade forward and drawn again with their own Philox keys (per-sample
              step parameters).  Equal ids (almost always) confirm the speculation.  Different ids roll that sample back: its
              tokens become the f32-grade ones after that update, its counter is set behind it, everything it did since is
              discarded (an epoch number per sample marks stale flags and queue entries), and it catches up in the following
              fast forwards.
  audit       a random share of the UNFLAGGED sample-updates goes through the same verification: `audit_rate` (2 %) until
              `audit_clean_target` (500) audits in a row were clean, `audit_rate_steady` (0.5 %) from then on; a mismatch starts
              the count again.  An audit whose ids differ is a certification miss: counted (`audit_mismatches`), repaired by the
              same roll-back, and its error raises the bounds.

  direct      speculation only pays while few sample-updates need the slow lane: a verified sample-update costs an f32-grade
              forward ON TOP of the fast one.  When more than `direct_share` (0.5) of the last two lanes' worth of sample-updates were
              flagged, the lane runs on the f32-grade engine alone (its draws need no certificate) until the share — still reported
              by the same kernel, from the f32-grade logits — has fallen below 0.6 x `direct_share`; so a certified call is never much
              slower than the f32-grade engine by itself.  That is the regime of the gibbs mode on weights whose output
              distributions are all nearly uniform (random initialisation): the entropies of the positions then lie ~4e-5 apart and
              the ORDER of two of them is decided below what f16 resolves (profiles/r06_certified_gibbs_*.txt).

The bounds come from the MEASURED error of this engine pair, not from a model of it: every verified item yields both engines'
logits for the same input; esmdiff_logit_error_stats reduces them per token row to the maximum and r.m.s. of the logit error e,
of the neighbouring-pair error d_v = e_v - e_(v+1), the row's range max e - min e, and the entropy error.  The pair that decides
a draw is arbitrary (winner and runner-up of a noisy race), so the bound in use is

    P = max(k_sigma * (largest per-row r.m.s. of d seen), max_factor * (largest row RANGE seen))          (eps = P / 2)
    2 E = max(k_sigma * sqrt(2) * (largest per-item r.m.s. of the entropy error seen),
              max_factor * (largest difference of the entropy errors of two rows of one sample seen))

widened by `boot_factor` (1.5) until `n_boot` (64) items have been seen; while NOTHING has been seen every sample-update is verified.  A
verified item whose range exceeds the P its update was certified with (or whose entropy error exceeds E) is a `violation`; it
raises the bound for everything that follows (also when eps is fixed).  The certificate is STATISTICAL: "k-sigma of the
measured error distribution + audit".  What it asserts is checked by measurement — job-level equality with the f32-grade
chain over many seeds (profiles/r06_certified_soak.txt: a rule-of-three bound) and the audit counters of every run — not by a
probability derived from a Gaussian model.

The result is "the f32-grade engine's chain unless an error exceeded the bounds on an unflagged, unaudited decision".  The
F32_SPLIT engine's logits do not depend on the batch a sample is evaluated in, so with it as `exact` the certified chain IS
that engine's chain, id for id (tests/test_gpu_strict.py).  No reference counterpart — the reference has one precision.
"""
from __future__ import annotations

import math
from collections import deque
from typing import List, Optional

import numpy as np
import torch

from .engine import Engine
from .schedule import DDPMSchedule
from .constants import STRUCTURE_MASK_TOKEN

_GAP_GRID = (0.25, 0.5, 1.0, 2.0, 4.0)      # stats["rerun_share_vs_eps"]: multiples of the bound in use
CERTIFICATE = "k-sigma statistical + audit"

# columns of the per-item statistics (one row per verified sample-update)
_ME, _SE, _MD, _SD, _ROWS, _SDROW, _RANGE, _MDH, _SDH, _RDH = range(10)
_NSTAT = 10


class _Async:
    """A small device -> host hand-over that does not stall the stream: pinned buffer + event (plain copy on CPU tensors)."""

    def __init__(self, t: torch.Tensor):
        if t.is_cuda:
            self.host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self.host.copy_(t, non_blocking=True)
            self.ev = torch.cuda.Event()
            self.ev.record()
        else:
            self.host, self.ev = t.clone(), None

    def get(self) -> np.ndarray:
        if self.ev is not None:
            self.ev.synchronize()
        return self.host.numpy()


class _DdpmRule:
    """model.py:543-607 as the loop sees it: T updates + the noise-removal pass; a sample without a MASK is complete."""
    mode, all_columns, track_mask, n_gaps = "ddpm", False, True, 1

    def __init__(self, sampler, schedule: DDPMSchedule, B: int, dev):
        fast, exact = sampler.fast, sampler.exact
        self.T = T = schedule.num_steps
        self.total = np.full(B, T + 1, dtype=np.int64)             # updates per sample; the last one is the noise removal
        tf_f, tf_e = fast.conditioning_rows(schedule.t_freq), exact.conditioning_rows(schedule.t_freq)
        self.tf = {id(fast): None if tf_f is None else tf_f.to(device=dev, dtype=torch.float32).contiguous(),
                   id(exact): None if tf_e is None else tf_e.to(device=dev, dtype=torch.float32).contiguous()}
        self.mc_t = np.array([float(schedule.mc_t[i]) for i in range(T)] + [0.0], dtype=np.float32)
        self.mc_s = np.array([float(schedule.mc_s[i]) for i in range(T)] + [0.0], dtype=np.float32)

    def skip_empty(self, step: np.ndarray) -> None:
        pass

    def cond(self, eng, steps: np.ndarray, steps_dev):
        tf = self.tf[id(eng)]
        if tf is None:
            return None
        return tf[int(steps[0])] if steps_dev is None else tf[steps_dev]

    def wants_steps_on_device(self, eng) -> bool:
        return self.tf[id(eng)] is not None

    def before_forward(self, eng, idx_dev) -> None:
        pass

    def params(self, eng, sample_ids: np.ndarray, steps: np.ndarray) -> np.ndarray:
        return eng.sample_step_params_host(sample_ids, self.mc_t[steps], self.mc_s[steps], steps, (steps == self.T).astype(np.int32))

    def draw(self, eng, x, seq, lg, par, seed, bounds=None, flags=None, gaps=None):
        if bounds is None:
            eng.ddpm_step_rows(x, lg, par, seed=seed)
        else:
            eng.ddpm_step_rows(x, lg, par, seed=seed, eps=0.5 * bounds[0], flags=flags, gaps=gaps.reshape(-1))

    def gap_units(self, bounds):
        return (bounds[0],)

    def shared_first_draw(self, eng, x, lg_one, seed, sample_offset, c0, n):
        raise NotImplementedError


class _GibbsRule:
    """esm's iterative_sampling_raw as the loop sees it: prompt s runs total[s] = min(num_steps, #masked) steps and unmasks
    table[t, s] positions at step t; no time conditioning; optional per-prompt backbone frames (geometric attention)."""
    mode, all_columns, track_mask, n_gaps = "gibbs", True, False, 2

    def __init__(self, sampler, table: torch.Tensor, temperature: float, top_p: float, frames, dev):
        self.table = table.detach().to("cpu", torch.int32).numpy()        # (T, B)
        T, B = self.table.shape
        nz = self.table > 0
        self.total = np.where(nz.any(0), T - np.argmax(nz[::-1], 0), 0).astype(np.int64)   # index of the last unmasking step + 1
        self.temperature, self.top_p = float(temperature), float(top_p)
        self.frames = None if frames is None else tuple(f.to(dev) for f in frames)
        self.engines = (sampler.fast, sampler.exact)

    def skip_empty(self, step: np.ndarray) -> None:
        """A step that unmasks nothing changes nothing (the plain loop still runs its forward; the ids cannot depend on it)."""
        T, B = self.table.shape
        for s in range(B):
            while step[s] < self.total[s] and self.table[step[s], s] <= 0:
                step[s] += 1

    def cond(self, eng, steps, steps_dev):
        return None

    def wants_steps_on_device(self, eng) -> bool:
        return False

    def before_forward(self, eng, idx_dev) -> None:
        if self.frames is not None:
            eng.set_frames(*(self.frames if idx_dev is None else tuple(f[idx_dev] for f in self.frames)))

    def params(self, eng, sample_ids: np.ndarray, steps: np.ndarray, local_ids: Optional[np.ndarray] = None) -> np.ndarray:
        ks = self.table[np.minimum(steps, self.table.shape[0] - 1), local_ids]
        return eng.gibbs_step_params_host(sample_ids, steps, np.where(steps < self.total[local_ids], ks, 0))

    def draw(self, eng, x, seq, lg, par, seed, bounds=None, flags=None, gaps=None):
        if bounds is None:
            eng.gibbs_step_rows(x, seq, lg, self.temperature, self.top_p, par, seed=seed)
        else:
            eng.gibbs_step_rows(x, seq, lg, self.temperature, self.top_p, par, seed=seed, pair_bound=bounds[0],
                                entropy_bound=bounds[1], flags=flags, gaps=gaps)

    def gap_units(self, bounds):
        return (bounds[0], 2.0 * bounds[1])

    def finish(self) -> None:
        if self.frames is not None:
            for eng in self.engines:
                eng.set_frames(None)


class CertifiedSampler:
    """fast: a reduced-precision Engine (f16 with the f32-grade head recommended: its logit error is 8x below bf16's, so 8x
    fewer close calls); exact: an f32-grade Engine of the same checkpoint (precision 'f32_split' or 'f32')."""

    def __init__(self, fast: Engine, exact: Engine, eps: Optional[float] = None, *, k_sigma: float = 6.0,
                 max_factor: float = 1.05, eps_floor: float = 1e-5, audit_rate: float = 0.02, verify_batch: int = 32,
                 n_boot: int = 64, boot_factor: float = 1.5, audit_seed: int = 0, audit_rate_steady: Optional[float] = None,
                 audit_clean_target: int = 500, entropy_eps: Optional[float] = None, direct_share: float = 0.5):
        if fast.device != exact.device:
            raise ValueError("both engines must live on the same GPU")
        if eps is not None and not eps > 0:
            raise ValueError("eps must be positive")
        if entropy_eps is not None and not entropy_eps > 0:
            raise ValueError("entropy_eps must be positive")
        if not k_sigma > 0 or not max_factor >= 1.0 or not boot_factor >= 1.0:
            raise ValueError("k_sigma must be positive, max_factor and boot_factor >= 1")
        if not 0.0 <= audit_rate <= 1.0 or (audit_rate_steady is not None and not 0.0 <= audit_rate_steady <= 1.0):
            raise ValueError("audit_rate must be in [0, 1]")
        if verify_batch < 1:
            raise ValueError("verify_batch must be >= 1")
        if not 0.0 < direct_share <= 1.0:
            raise ValueError("direct_share must be in (0, 1]")
        self.direct_share = float(direct_share)      # see _run: above this share of verified sample-updates the exact engine draws itself
        self.fast, self.exact = fast, exact
        self.eps = None if eps is None else float(eps)          # None: from the error distribution (module docstring)
        self.entropy_eps = None if entropy_eps is None else float(entropy_eps)
        self.k_sigma, self.max_factor, self.eps_floor = float(k_sigma), float(max_factor), float(eps_floor)
        self.audit_rate, self.verify_batch = float(audit_rate), int(verify_batch)
        # the steady rate applies once audit_clean_target audits in a row were clean (all calls of this engine pair)
        self.audit_rate_steady = min(self.audit_rate, 0.005) if audit_rate_steady is None else float(audit_rate_steady)
        self.audit_clean_target, self.audit_clean_run = int(audit_clean_target), 0
        self.n_boot, self.boot_factor = int(n_boot), float(boot_factor)
        self._audit_rng = np.random.default_rng(audit_seed)
        # running error estimate of this engine pair (all calls, both modes): per-item r.m.s. and maxima
        self.sigma_d_seen = 0.0      # largest per-ROW r.m.s. of the neighbouring-pair error d (what k_sigma multiplies)
        self.sigma_d_item_seen = 0.0 # largest per-item (sample-update) r.m.s. of d, for the reports
        self.sigma_e_seen = 0.0      # largest per-item r.m.s. of the logit error e
        self.max_d_seen = 0.0
        self.range_seen = 0.0        # largest per-row max e - min e: the bound on the error of ANY pair's difference
        self.err_seen = 0.0          # largest |e| (r04's statistic, kept for the reports)
        self.sigma_h_seen = 0.0      # largest per-item r.m.s. of the entropy error
        self.max_dh_seen = 0.0       # largest entropy error
        self.range_dh_seen = 0.0     # largest difference of the entropy errors of two rows of one sample (what decides their order)
        self.n_seen = 0              # verified items the estimate rests on
        self.trace_sample: Optional[int] = None   # diagnostics: stats["trace"] lists every event of this (local) sample index
        self.lane_memory: dict = {}  # mode -> did the last call's first window of sample-updates exceed direct_share? (start there)
        self.pair_raise = 0.0        # pair bound forced by violations (also with a fixed eps)
        self.entropy_raise = 0.0
        self.stats: dict = {}

    # ---- bounds -----------------------------------------------------------------------------------------------------------
    def pair_bound(self) -> float:
        """P: the bound on the error of a logit difference the next update is certified with (= 2 eps)."""
        if self.eps is not None:
            return max(2.0 * self.eps, self.pair_raise)
        p = max(self.k_sigma * self.sigma_d_seen, self.max_factor * self.range_seen)
        if self.n_seen < self.n_boot:
            p *= self.boot_factor
        return max(p, self.pair_raise, 2.0 * self.eps_floor)

    def entropy_bound(self) -> float:
        """E: half of the bound on the error of the DIFFERENCE of two rows' entropies (gibbs mode: the order of the positions; the
        kernel asks for 2 E between the last selected and the first unselected one): k_sigma x the r.m.s. of a difference of two
        independent entropy errors, or max_factor x the largest difference seen between two rows of one sample."""
        if self.entropy_eps is not None:
            return max(self.entropy_eps, self.entropy_raise)
        e = 0.5 * max(self.k_sigma * math.sqrt(2.0) * self.sigma_h_seen, self.max_factor * self.range_dh_seen)
        if self.n_seen < self.n_boot:
            e *= self.boot_factor
        return max(e, self.entropy_raise, 1e-7)

    def _eps_now(self) -> float:
        return 0.5 * self.pair_bound()

    def _audit_rate_now(self) -> float:
        return self.audit_rate_steady if self.audit_clean_run >= self.audit_clean_target else self.audit_rate

    def _blind(self, gibbs: bool = False) -> bool:
        """No verified item yet and a bound is to come from them: nothing can be certified, every sample-update is verified."""
        return self.n_seen == 0 and (self.eps is None or (gibbs and self.entropy_eps is None))

    def _observe_items(self, st: np.ndarray, V: int) -> None:
        """st (n, >= 10): per item max |e|, sum e^2, max |d|, sum d^2, masked rows, largest per-row sum d^2, largest row range,
        largest |entropy error|, sum of its squares, range of the entropy error over the item's masked rows."""
        for it in st:
            rows = it[_ROWS]
            if rows <= 0:
                continue
            self.n_seen += 1
            self.err_seen = max(self.err_seen, float(it[_ME]))
            self.max_d_seen = max(self.max_d_seen, float(it[_MD]))
            self.range_seen = max(self.range_seen, float(it[_RANGE]))
            self.sigma_e_seen = max(self.sigma_e_seen, math.sqrt(float(it[_SE]) / (rows * (V - 1))))
            self.sigma_d_item_seen = max(self.sigma_d_item_seen, math.sqrt(float(it[_SD]) / (rows * (V - 2))))
            self.sigma_d_seen = max(self.sigma_d_seen, math.sqrt(float(it[_SDROW]) / (V - 2)))
            self.max_dh_seen = max(self.max_dh_seen, float(it[_MDH]))
            self.range_dh_seen = max(self.range_dh_seen, float(it[_RDH]))
            self.sigma_h_seen = max(self.sigma_h_seen, math.sqrt(float(it[_SDH]) / rows))

    def _item_stats(self, lg_fast: torch.Tensor, lg_exact: torch.Tensor, x_in: torch.Tensor, all_columns: bool) -> torch.Tensor:
        """(n, 10) on the device, per sample over its masked rows (column order: _ME .. _RDH)."""
        st = self.exact.logit_error_stats(lg_fast, lg_exact, x_in, all_columns)
        m = x_in == STRUCTURE_MASK_TOKEN
        rows = m.sum(1).to(torch.float32)
        dh = st[..., 5]
        inf = torch.full_like(dh, float("inf"))
        rdh = (torch.where(m, dh, -inf).amax(1) - torch.where(m, dh, inf).amin(1)).clamp_min(0.0).nan_to_num(0.0, 0.0, 0.0)
        return torch.stack([st[..., 0].amax(1), st[..., 1].sum(1), st[..., 2].amax(1), st[..., 3].sum(1), rows, st[..., 3].amax(1),
                            st[..., 4].amax(1), dh.abs().amax(1), (dh * dh).sum(1), rdh], 1)

    # ---- the two entry points ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def ddpm_sample(self, sequence_tokens: torch.Tensor, schedule: DDPMSchedule, *, seed: int, sample_offset: int = 0,
                    input_prior: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same arguments and result layout as Engine.ddpm_sample.  self.stats describes the call.
        The batch may hold MORE samples than fast.max_batch (the CLI's batch loop, sample_esmdiff.py:181-216, as one stream): every
        fast forward takes the fast.max_batch unfinished samples with the lowest indices, so a sample that was rolled back
        rides along with later ones instead of leaving the GPU with a tail of one-sample forwards.  Ids do not depend on the
        lane width (Philox is keyed by the global sample index)."""
        B, L = sequence_tokens.shape
        dev = self.fast.device
        seq = sequence_tokens.to(device=dev, dtype=torch.int64).contiguous()
        if input_prior is None:
            x = torch.full((B, L), STRUCTURE_MASK_TOKEN, dtype=torch.int64, device=dev)
        else:
            if tuple(input_prior.shape) != (B, L):
                raise ValueError(f"Invalid input_prior shape: {tuple(input_prior.shape)} v.s. (seq) {(B, L)}")
            x = input_prior.to(device=dev, dtype=torch.int64).contiguous().clone()
        rule = _DdpmRule(self, schedule, B, dev)
        identical = input_prior is None and B > 1 and schedule.num_steps > 0 and bool((seq == seq[:1]).all())
        return self._run(rule, seq, x, seed, sample_offset, identical, known_masks=input_prior is not None or schedule.num_steps == 0)

    @torch.no_grad()
    def gibbs_sample(self, sequence_tokens: torch.Tensor, x0: torch.Tensor, n_unmask_table: torch.Tensor, temperature: float,
                     top_p: float, *, seed: int, sample_offset: int = 0, frames=None) -> torch.Tensor:
        """Same arguments and result layout as Engine.gibbs_sample (n_unmask_table (T, B) int32: positions to unmask per step and
        prompt), plus `frames` = (rot (B, L, 3, 3), trans (B, L, 3), has_frame (B, L)) for coordinate conditioning (what
        Engine.set_frames takes; the sampler hands every forward the frames of the prompts it runs).  GenerationConfig options
        (strategy, invalid_ids) must have been set on BOTH engines.  self.stats describes the call; more prompts than
        fast.max_batch are streamed as in ddpm_sample."""
        B, L = sequence_tokens.shape
        dev = self.fast.device
        seq = sequence_tokens.to(device=dev, dtype=torch.int64).contiguous()
        if tuple(x0.shape) != (B, L):
            raise ValueError(f"Invalid x0 shape: {tuple(x0.shape)} v.s. (seq) {(B, L)}")
        x = x0.to(device=dev, dtype=torch.int64).contiguous().clone()
        tab = torch.as_tensor(n_unmask_table)
        if tab.dim() != 2 or tab.shape[1] != B:
            raise ValueError(f"n_unmask_table must be (T, B = {B}), got {tuple(tab.shape)}")
        rule = _GibbsRule(self, tab, temperature, top_p, frames, dev)
        identical = B > 1 and bool((seq == seq[:1]).all()) and bool((x == x[:1]).all()) and bool((tab == tab[:, :1]).all()) and \
            (rule.frames is None or all(bool((f == f[:1]).all()) for f in rule.frames))
        try:
            return self._run(rule, seq, x, seed, sample_offset, identical, known_masks=True)
        finally:
            rule.finish()

    # ---- the loop ---------------------------------------------------------------------------------------------------------
    def _run(self, rule, seq: torch.Tensor, x: torch.Tensor, seed: int, sample_offset: int, identical: bool,
             known_masks: bool) -> torch.Tensor:
        import time
        fast, exact = self.fast, self.exact
        B, L = seq.shape
        W = min(B, fast.max_batch)                                 # lane width: samples per fast forward
        t_call = time.perf_counter()
        dev = fast.device
        MASK = STRUCTURE_MASK_TOKEN
        gibbs = rule.mode == "gibbs"
        if hasattr(fast, "_check_ids"):
            fast._check_ids(seq, x)                                # once: every later forward sees these ids or the kernels' own
        V = exact.cfg.n_structure_heads
        total = rule.total                                         # updates per sample
        Tmax = int(total.max()) if B else 0
        NG = rule.n_gaps

        ld = fast.ld_logits
        lbuf = [torch.empty(W, L, ld, dtype=torch.float32, device=dev) for _ in range(2)]     # fast logits, two updates in flight
        cap = max(1, min(exact.max_batch, max(2 * self.verify_batch, 64)))   # a full pool is verified at once, whatever verify_batch
        pools = [{"before": torch.empty(cap, L, dtype=torch.int64, device=dev), "after": torch.empty(cap, L, dtype=torch.int64, device=dev),
                  "lg": torch.empty(cap, L, ld, dtype=torch.float32, device=dev), "items": []} for _ in range(2)]
        fill = 0                                                   # index of the pool that is filling

        step = np.zeros(B, dtype=np.int64)                         # next update of each sample (total[s] = done)
        epoch = np.zeros(B, dtype=np.int64)
        mask_known = np.zeros(B, dtype=bool)                       # ddpm: is has_mask[s] valid for the sample's current state?
        has_mask = np.ones(B, dtype=bool)
        updates: deque = deque()                                   # fast updates whose flags are still on their way
        verifies: deque = deque()
        st = {"mode": rule.mode, "certificate": CERTIFICATE, "samples": B, "updates": Tmax, "fast_launches": 0, "verify_launches": 0,
              "verify_batch_sizes": [], "flagged": 0, "flag_reasons": {"race": 0, "nucleus": 0, "order": 0, "no_estimate": 0},
              "audit_checked": 0, "audit_mismatches": 0, "audit_eps_violations": 0,
              "audit_max_logit_err": 0.0, "audit_max_pair_err": 0.0, "audit_max_range_err": 0.0, "corrections": 0, "eps_violations": 0,
              "entropy_violations": 0, "rollback_updates_discarded": 0, "sample_forwards_fast": 0, "sample_forwards_exact": 0,
              "max_logit_err_observed": 0.0, "max_pair_err_observed": 0.0, "max_range_err_observed": 0.0,
              "max_entropy_err_observed": 0.0, "flagged_per_update": [0] * max(Tmax, 1), "eps_used": [], "entropy_eps_used": [],
              "direct_lane_from_launch": None, "direct_lane_share_seen": None, "sample_forwards_direct": 0, "direct_launches": 0,
              "direct_lane_switches": []}
        recent: deque = deque(maxlen=2 * max(W, 16))               # 1 / 0 per live sample-update that drew: did it need the slow lane?
        # the lane runs on the f32-grade engine alone; a call starts where the last call of this mode started out (the CLI runs
        # batch after batch of one protein): the direct lane keeps reporting the share, so it leaves as soon as speculation pays
        direct = [bool(self.lane_memory.get(rule.mode, False))]
        first_share: list = []
        gap_log: List[List[float]] = [[] for _ in range(NG)]       # smallest gaps of every live sample-update / the bound they ran with
        timers: list = []                                          # (lane, start event, end event) of every forward + draw

        def runnable() -> np.ndarray:
            if rule.track_mask:      # ddpm: the last update (noise removal) only for samples known to hold a MASK
                return (step < total - 1) | ((step == total - 1) & mask_known & has_mask)
            return step < total

        def settle_final(s: int) -> None:
            # ddpm: a sample without a MASK is carried through every remaining update and through the noise removal unchanged
            # (model.py:530-532, 575-579, 606-607): nothing left to run for it
            if rule.track_mask and mask_known[s] and not has_mask[s]:
                step[s] = total[s]

        if rule.track_mask and known_masks:
            has_mask[:], mask_known[:] = (x == MASK).any(1).cpu().numpy(), True
            for s in range(B):
                settle_final(s)
        rule.skip_empty(step)

        def up(a: np.ndarray) -> torch.Tensor:
            """Host array -> device through pinned memory: never waits for the work already queued on the stream (a pageable
            copy is performed synchronously BEHIND it, which would park the host for the length of a forward)."""
            t = torch.from_numpy(a)
            return t.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else t

        def tick():
            if dev.type != "cuda":
                return None
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev

        def params(eng, local: np.ndarray, steps: np.ndarray) -> np.ndarray:
            if gibbs:
                return rule.params(eng, sample_offset + local, steps, local)
            return rule.params(eng, sample_offset + local, steps)

        # -- the error estimate of a new engine pair starts from a probe of samples that have something to decide
        if self._blind(gibbs) and B:
            cand = np.nonzero(step < total)[0]
            cand = cand[:1] if identical else cand[:2]
            if len(cand):
                idx_d = up(cand)
                xs, sq = x[idx_d], seq[idx_d]
                steps0 = step[cand].copy()
                outs = []
                for eng in (exact, fast):
                    rule.before_forward(eng, idx_d)
                    sd = up(steps0) if (rule.wants_steps_on_device(eng) and len(cand) > 1) else None
                    outs.append(eng.forward_logits(xs, sq, rule.cond(eng, steps0, sd)).clone())
                self._observe_items(self._item_stats(outs[1], outs[0], xs, rule.all_columns).cpu().numpy(), V)
                st["sample_forwards_fast"] += len(cand)
                st["sample_forwards_exact"] += len(cand)

        # -- identical inputs at the first update (the CLI repeats ONE protein): the fast forward of that update runs on a few
        #    samples and serves all of them — each draws with its own noise, is flagged by the same rule and verified like any other
        step_first = step.copy()
        n_share = min(W, int(getattr(fast, "shared_forward_batch", lambda b, l: 1)(W, L))) if identical else 0
        shared_lg: list = []                                       # the first update's logits of ONE sample, once computed
        shared_lg_exact: list = []

        def launch_direct(active: np.ndarray) -> None:
            """The lane on the f32-grade engine itself: plain draws, nothing to flag, verify or audit."""
            n = len(active)
            steps = step[active].copy()
            par = up(params(exact, active, steps))
            mixed = not (steps == steps[0]).all()
            steps_d = up(steps) if (mixed and rule.wants_steps_on_device(exact)) else None
            idx_d = up(active)
            xa, sa = x[idx_d], seq[idx_d]
            bounds = (self.pair_bound(), self.entropy_bound())
            e0 = tick()
            first = n_share > 0 and not mixed and bool((steps == step_first[active]).all()) and bool((epoch[active] == 0).all())
            if first and n > 1:                                    # identical inputs: one sample's forward serves all (f32-grade logits do
                if not shared_lg_exact:                            # not depend on the batch they are computed in)
                    rule.before_forward(exact, up(active[:1]))
                    shared_lg_exact.append(exact.forward_logits(xa[:1], sa[:1], rule.cond(exact, steps[:1], None), check_ids=False).clone())
                    st["sample_forwards_exact"] += 1 - n
                else:
                    st["sample_forwards_exact"] -= n
                lg = shared_lg_exact[0].expand(n, L, shared_lg_exact[0].shape[-1]).contiguous()
            else:
                rule.before_forward(exact, idx_d)
                lg = exact.forward_logits(xa, sa, rule.cond(exact, steps, steps_d), check_ids=False)
            # the report is taken here too (same ids, bit for bit): it says what the fast lane WOULD have had to verify, which is
            # what decides when speculation pays again
            flags = torch.zeros(n, dtype=torch.int32, device=dev)
            gaps = torch.full((n, NG), float("inf"), dtype=torch.float32, device=dev)
            rule.draw(exact, xa, sa, lg, par, seed, bounds, flags, gaps)
            timers.append(("direct", e0, tick()))
            x[idx_d] = xa
            back = _Async(torch.cat([flags.to(torch.float32)[:, None], gaps, (xa == MASK).any(1).to(torch.float32)[:, None]], 1))
            updates.append({"active": active, "steps": steps, "epochs": epoch[active].copy(), "back": back, "direct": True})
            step[active] += 1
            mask_known[active] = False
            rule.skip_empty(step)
            st["sample_forwards_direct"] += n
            st["sample_forwards_exact"] += n
            st["direct_launches"] += 1
            st["eps_used"].append(0.5 * bounds[0])                # (the bounds the report of this launch was taken with)
            st["entropy_eps_used"].append(bounds[1])

        def launch_fast(active: np.ndarray, which: int) -> None:
            if direct[0]:
                return launch_direct(active[:exact.max_batch])
            n = len(active)
            full = n == B
            steps = step[active].copy()
            # every upload first: nothing below waits for the host
            par = up(params(fast, active, steps))
            mixed = not (steps == steps[0]).all()
            steps_d = up(steps) if (mixed and rule.wants_steps_on_device(fast)) else None
            idx_d = None if full else up(active)
            if full:
                xa, sa = x, seq
            else:
                xa, sa = x[idx_d], seq[idx_d]
            prev = xa.clone()
            bounds = (self.pair_bound(), self.entropy_bound())
            e0 = tick()
            first = n_share > 0 and not mixed and bool((steps == step_first[active]).all()) and bool((epoch[active] == 0).all())
            if first and (shared_lg or n > n_share):               # the shared first update: every input row is the same
                n_fwd = 0
                if not shared_lg:
                    rule.before_forward(fast, None if n_share == B else up(active[:n_share]))
                    lg1 = fast.forward_logits(xa[:n_share], sa[:n_share], rule.cond(fast, steps[:1], None), check_ids=False)
                    shared_lg.append(lg1[:1].clone())
                    n_fwd = n_share
                lg = lbuf[which][:n, :, :shared_lg[0].shape[-1]]
                lg.copy_(shared_lg[0].expand(n, L, shared_lg[0].shape[-1]))
            else:
                rule.before_forward(fast, idx_d)
                lg = fast.forward_logits(xa, sa, rule.cond(fast, steps, steps_d), out=lbuf[which][:n], check_ids=False)
                n_fwd = n
            flags = torch.zeros(n, dtype=torch.int32, device=dev)
            gaps = torch.full((n, NG), float("inf"), dtype=torch.float32, device=dev)
            rule.draw(fast, xa, sa, lg, par, seed, bounds, flags, gaps)
            timers.append(("fast", e0, tick()))
            if not full:
                x[idx_d] = xa
            after = xa.clone() if full else xa
            back = _Async(torch.cat([flags.to(torch.float32)[:, None], gaps, (after == MASK).any(1).to(torch.float32)[:, None]], 1))
            updates.append({"active": active, "steps": steps, "epochs": epoch[active].copy(), "prev": prev, "after": after,
                            "lg": lbuf[which], "back": back, "bounds": bounds, "blind": self._blind(gibbs)})
            step[active] += 1
            mask_known[active] = False
            rule.skip_empty(step)
            st["fast_launches"] += 1
            st["sample_forwards_fast"] += n_fwd
            st["eps_used"].append(0.5 * bounds[0])
            st["entropy_eps_used"].append(bounds[1])

        def process_update(rec) -> None:
            nonlocal fill
            res = rec["back"].get()
            fl, gp, hm = res[:, 0], res[:, 1:1 + NG], res[:, 1 + NG]
            live = epoch[rec["active"]] == rec["epochs"]
            if rec.get("direct"):
                for j in np.nonzero(live)[0]:
                    s = int(rec["active"][j])
                    has_mask[s], mask_known[s] = bool(hm[j] > 0), True
                    settle_final(s)
                    if np.isfinite(gp[j, 0]):
                        recent.append(1 if fl[j] > 0 else 0)
                return
            units = rule.gap_units(rec["bounds"])
            pick, kinds = [], []
            audits_wanted = 2 if (self.eps is None and self.n_seen < self.n_boot) else 0
            for j in np.nonzero(live)[0]:
                s, k = int(rec["active"][j]), int(rec["steps"][j])
                has_mask[s], mask_known[s] = bool(hm[j] > 0), True
                settle_final(s)
                drew = bool(np.isfinite(gp[j, 0]))                 # (a sample-update that drew nothing has nothing to verify)
                for c in range(NG):
                    if np.isfinite(gp[j, c]) and units[c] > 0:
                        gap_log[c].append(float(gp[j, c]) / units[c])
                f = int(fl[j])
                if s == self.trace_sample:
                    st.setdefault("trace", []).append(("update", k, "flags", f, "gaps", [float(v) for v in gp[j]], "epoch", int(rec["epochs"][j]),
                                                       "launch", st["fast_launches"], "bounds", rec["bounds"]))
                if drew:
                    recent.append(1 if f > 0 else 0)
                if f > 0 or (rec["blind"] and drew):
                    pick.append(j); kinds.append("flag")
                    st["flagged"] += 1
                    st["flagged_per_update"][min(k, len(st["flagged_per_update"]) - 1)] += 1
                    if f & 1: st["flag_reasons"]["race"] += 1
                    if f & 2: st["flag_reasons"]["nucleus"] += 1
                    if f & 4: st["flag_reasons"]["order"] += 1
                    if f == 0: st["flag_reasons"]["no_estimate"] += 1
                elif drew and (self._audit_rng.random() < self._audit_rate_now() or audits_wanted > 0):
                    pick.append(j); kinds.append("audit")
                    audits_wanted -= 1
            while pick:
                pool = pools[fill]
                room = cap - len(pool["items"])
                if room == 0:
                    launch_verify()
                    continue
                take, pick = pick[:room], pick[room:]
                tk, kinds = kinds[:room], kinds[room:]
                lo = len(pool["items"])
                jd = up(np.asarray(take, dtype=np.int64))
                pool["before"][lo:lo + len(take)] = rec["prev"][jd]
                pool["after"][lo:lo + len(take)] = rec["after"][jd]
                pool["lg"][lo:lo + len(take)] = rec["lg"][jd]
                for j, kind in zip(take, tk):
                    pool["items"].append({"s": int(rec["active"][j]), "k": int(rec["steps"][j]), "epoch": int(rec["epochs"][j]),
                                          "kind": kind, "bounds": rec["bounds"]})

        def launch_verify() -> None:
            nonlocal fill
            pool = pools[fill]
            n = len(pool["items"])
            if n == 0:
                return
            while verifies:                                       # settle the earlier batch first: its roll-backs make entries stale
                process_verify(verifies.popleft())
            items = pool["items"]
            ss = np.array([it["s"] for it in items], dtype=np.int64)
            ks = np.array([it["k"] for it in items], dtype=np.int64)
            par = up(params(exact, ss, ks))
            ss_d = up(ss)
            ks_d = up(ks) if (rule.wants_steps_on_device(exact) and n > 1) else None
            xs = pool["before"][:n].clone()
            sq = seq[ss_d]
            e0 = tick()
            rule.before_forward(exact, ss_d)
            lg2 = exact.forward_logits(xs, sq, rule.cond(exact, ks, ks_d), check_ids=False)
            stats = self._item_stats(pool["lg"][:n], lg2, pool["before"][:n], rule.all_columns)
            rule.draw(exact, xs, sq, lg2, par, seed)
            timers.append(("verify", e0, tick()))
            neq = (xs != pool["after"][:n]).any(1).to(torch.float32)
            hm = (xs == MASK).any(1).to(torch.float32)
            back = _Async(torch.cat([stats, neq[:, None], hm[:, None]], 1))
            verifies.append({"items": items, "xs": xs, "back": back})
            pool["items"] = []
            fill ^= 1
            st["verify_launches"] += 1
            st["verify_batch_sizes"].append(n)
            st["sample_forwards_exact"] += n

        def process_verify(rec) -> None:
            res = rec["back"].get()
            self._observe_items(res[:, :_NSTAT], V)
            for j, it in enumerate(rec["items"]):
                me, md, rg, dh, rdh = float(res[j, _ME]), float(res[j, _MD]), float(res[j, _RANGE]), float(res[j, _MDH]), float(res[j, _RDH])
                neq, hm = res[j, _NSTAT] > 0, res[j, _NSTAT + 1] > 0
                st["max_logit_err_observed"] = max(st["max_logit_err_observed"], me)
                st["max_pair_err_observed"] = max(st["max_pair_err_observed"], md)
                st["max_range_err_observed"] = max(st["max_range_err_observed"], rg)
                st["max_entropy_err_observed"] = max(st["max_entropy_err_observed"], dh)
                audit = it["kind"] == "audit"
                if audit:
                    st["audit_checked"] += 1
                    st["audit_max_logit_err"] = max(st["audit_max_logit_err"], me)
                    st["audit_max_pair_err"] = max(st["audit_max_pair_err"], md)
                    st["audit_max_range_err"] = max(st["audit_max_range_err"], rg)
                P_used, E_used = it["bounds"]
                if rg > P_used:                                  # some pair's error left the bound this update was certified with
                    st["eps_violations"] += 1
                    st["audit_eps_violations"] += int(audit)
                    self.pair_raise = max(self.pair_raise, self.max_factor * rg)
                if gibbs and rdh > 2.0 * E_used:                 # two rows' entropy errors were further apart than the order test allowed for
                    st["entropy_violations"] += 1
                    self.entropy_raise = max(self.entropy_raise, 0.5 * self.max_factor * rdh)
                s = it["s"]
                stale = epoch[s] != it["epoch"]
                if s == self.trace_sample:
                    st.setdefault("trace", []).append(("verify", it["k"], it["kind"], "neq", bool(neq), "stale", bool(stale), "epoch", it["epoch"],
                                                       "now", int(epoch[s]), "step_now", int(step[s])))
                if audit and not stale:
                    self.audit_clean_run = 0 if neq else self.audit_clean_run + 1
                if stale or not neq:
                    continue                                      # stale (the sample was rolled back meanwhile) or confirmed
                st["audit_mismatches" if audit else "corrections"] += 1
                if audit:                                         # a miss of the certificate: widen the bounds for what follows
                    self.pair_raise = max(self.pair_raise, self.max_factor * max(rg, P_used) * 1.5)
                    if gibbs:
                        self.entropy_raise = max(self.entropy_raise, self.max_factor * max(0.5 * rdh, E_used) * 1.5)
                st["rollback_updates_discarded"] += int(step[s] - (it["k"] + 1))
                x[s] = rec["xs"][j]
                step[s] = it["k"] + 1
                epoch[s] += 1
                has_mask[s], mask_known[s] = bool(hm), True
                settle_final(s)
                rule.skip_empty(step)

        which = 0
        t_tail = None
        while True:
            active = np.nonzero(runnable())[0][:W]
            if t_tail is None and len(active) < W and st["fast_launches"] > 0:
                if fast.device.type == "cuda":                    # the lane is no longer full: what follows is the tail
                    torch.cuda.synchronize(fast.device)
                t_tail = time.perf_counter()
            if len(active):
                launch_fast(active, which)
                which ^= 1
            # results one iteration late: while the host waits for them the GPU is already inside the forward just launched
            while len(updates) > (1 if len(active) else 0):
                process_update(updates.popleft())
            if len(recent) >= max(W, 16):
                share = sum(recent) / len(recent)
                if not first_share:
                    first_share.append(share)
                    self.lane_memory[rule.mode] = share > self.direct_share
                if not direct[0] and share > self.direct_share:
                    direct[0] = True                              # too many open decisions for speculation to pay (class docstring)
                    st["direct_lane_switches"].append(("direct", st["fast_launches"] + st["direct_launches"], round(share, 4)))
                    if st["direct_lane_from_launch"] is None:
                        st["direct_lane_from_launch"], st["direct_lane_share_seen"] = st["fast_launches"], round(share, 4)
                elif direct[0] and share < 0.6 * self.direct_share:
                    direct[0] = False                             # the chain reached a phase where few decisions are open
                    st["direct_lane_switches"].append(("fast", st["fast_launches"] + st["direct_launches"], round(share, 4)))
            while verifies:                                       # (launched in an earlier iteration, or nothing else to do)
                process_verify(verifies.popleft())
            queued = len(pools[fill]["items"])
            if queued >= self.verify_batch or (queued and not len(active)) or (queued and self._blind(gibbs)):
                launch_verify()
            if not len(active) and not updates and not verifies and not pools[fill]["items"]:
                if (step >= total).all():
                    break
                if not runnable().any():
                    raise RuntimeError("certified sampler stalled: unfinished samples without pending work")   # (unreachable)

        if fast.device.type == "cuda":
            torch.cuda.synchronize(fast.device)
        st["seconds"] = round(time.perf_counter() - t_call, 4)
        st["tail_seconds"] = 0.0 if t_tail is None else round(time.perf_counter() - t_tail, 4)     # after the last full-lane forward
        st["lane_width"] = W
        if timers and timers[0][1] is not None:                   # device time inside the two lanes (HIP events on the stream)
            for lane in ("fast", "verify", "direct"):
                st[f"gpu_seconds_{lane}"] = round(sum(a.elapsed_time(b) for ln_, a, b in timers if ln_ == lane) * 1e-3, 4)
        gl = np.array(gap_log[0]) if gap_log[0] else np.zeros(0)
        n_upd = max(1, len(gl))
        used, used_h = st.pop("eps_used"), st.pop("entropy_eps_used")
        st.update({
            "eps": self.eps if self.eps is not None else "auto", "k_sigma": self.k_sigma, "max_factor": self.max_factor,
            "audit_rate": self.audit_rate, "audit_rate_steady": self.audit_rate_steady, "audit_rate_now": self._audit_rate_now(),
            "audit_clean_run": self.audit_clean_run, "verify_batch": self.verify_batch,
            "eps_min_used": min(used) if used else None, "eps_max_used": max(used) if used else None,
            "sigma_pair_err": self.sigma_d_seen, "sigma_pair_err_per_item": self.sigma_d_item_seen, "sigma_logit_err": self.sigma_e_seen,
            "max_pair_err_all_calls": self.max_d_seen, "max_range_err_all_calls": self.range_seen,
            "max_logit_err_all_calls": self.err_seen, "items_seen_all_calls": self.n_seen, "first_update_shared": bool(identical),
            # share of sample-updates whose smallest gap is within m x the bound in use: what a bound of m x that one would re-run
            "rerun_share_vs_eps": {str(m): round(float((gl <= m).sum()) / n_upd, 5) for m in _GAP_GRID},
            "rerun_share": round(st["flagged"] / n_upd, 5), "direct_share": self.direct_share,
        })
        if gibbs:
            gh = np.array(gap_log[1]) if gap_log[1] else np.zeros(0)
            st.update({"entropy_eps_min_used": min(used_h) if used_h else None, "entropy_eps_max_used": max(used_h) if used_h else None,
                       "sigma_entropy_err": self.sigma_h_seen, "max_entropy_err_all_calls": self.max_dh_seen,
                       "max_entropy_err_difference_all_calls": self.range_dh_seen,
                       "order_share_vs_bound": {str(m): round(float((gh <= m).sum()) / max(1, len(gh)), 5) for m in _GAP_GRID}})
        self.stats = st
        return x
