"""Certified sampling: the ids of the f32-grade chain at close to the reduced-precision engine's speed.

The state of the ancestral sampler (model.py:543-581) between two updates is a matrix of token ids, so a rounding error of
the network does not accumulate from update to update: as long as every draw of an update came out the same, the next
update starts from the identical input.  A draw is an arg-max of q_v / g_v (model.py:24-28) with q_v = exp(z_v - lse) *
(mc_t - mc_s), or mc_s for the mask column.  Between two tokens the logsumexp cancels, so the log of the ratio of two
candidates moves by the error of the logit DIFFERENCE z_a - z_b; against the mask column it moves by the error of
z_a - lse, and lse is a convex combination of the logits (the common part of the error cancels there too).  If that pair
error is at most P = 2 eps, a winner that beats the runner-up by more than the factor exp(2 eps) is the winner for the
f32-grade logits as well.  The sampler kernel reports, per sample, whether all of its draws were that clear
(esmdiff_ddpm_step_rows, csrc/sampler.hip).

What happens to a sample with a close call (r05: speculative, batched, audited).  The samples of a batch are independent
(model.py:583-607 is row-wise), so nothing forces a flagged sample to be settled in the update it was flagged in:

  fast lane   every sample carries its own update counter; one reduced-precision forward per iteration advances all unfinished
              samples (one sigma per sample, esmdiff_forward_logits_sigmas), flagged or not — a flagged sample CONTINUES on its
              fast ids, speculatively.  Nothing in the lane waits for the host: flags come back through pinned memory one
              iteration late, while the GPU is already inside the next forward.
  slow lane   a flagged sample-update (tokens before, fast tokens after, fast logits) is queued; when `verify_batch` of them
              have gathered they are evaluated in ONE f32-grade forward — each at its own sigma — and drawn again with their own
              Philox keys (per-sample step parameters).  Equal ids (almost always: the flag says "could differ", ~1 in 12
              does) confirm the speculation.  Different ids roll that sample back: its tokens become the f32-grade ones after
              that update, its counter is set behind it, everything it did since is discarded (an epoch number per sample
              marks stale flags and queue entries), and it catches up in the following fast forwards.
  audit       a random `audit_rate` of the UNFLAGGED sample-updates goes through the same verification.  An audit whose ids
              differ is a certification miss: counted (`audit_mismatches`), repaired by the same roll-back, and its error
              raises eps.  stats carries audit_checked / audit_mismatches / audit_max_logit_err / audit_max_pair_err.

`eps` (eps=None, the default) is chosen from the error DISTRIBUTION, not from a maximum: every verified item yields both
engines' logits for the same input, esmdiff_logit_error_stats reduces them per token row to the r.m.s. and the maximum of the
logit error e_v and of the pair error d_v = e_v - e_(v+1) (4 099 pairs per masked row).  Within one row the errors are the
projections of ONE hidden-state error onto 4 101 head rows — Gaussian-like with a scale that belongs to the row — so the
statistic is the per-row r.m.s., and the bound takes the largest one seen (a pooled r.m.s. is a scale mixture: on weights with
trained statistics its tail reached 7 sigma, per row it stays at 5-5.5).  The pair bound in use is

    P = max(k_sigma * (largest per-row r.m.s. of d seen so far), max_factor * (largest |d| seen so far)),   eps = P / 2

Measured at configs[1] (f16 body + f32-grade head against F32_SPLIT): sigma_d = 3.1e-4, the largest |d| of ~1e6 pairs per item
sits at 5.3-5.7 sigma, as a Gaussian's does.  With k_sigma = 6 on the LARGEST row sigma (3.6e-4; the typical row's is 15 % smaller)
and Gaussian pair errors a draw whose gap is just outside the flagged band flips with probability <= 2 Q(6) = 2e-9; integrated
over the gap density rho (0.13 per unit log gap and masked row, measured: stats["rerun_share_vs_eps"]) the miss probability is
<= 4 rho sigma_d phi(k) / k^2 ~ 3e-14 per draw, ~1e-8 per 100-sample job of 335 000 draws — and the audit watches the assumption.  (r04 used 2 x the largest |e| = ~16 sigma_d: 2.5x the
re-runs for no measurable gain in safety; its rule is still available as a fixed `eps`.)  Until `n_boot` items have been seen the bound is
widened by `boot_factor` and at least two samples per update are audited; the very first call starts from a probe of two samples
on both engines.  A verified item whose largest |d| exceeds the P its update was certified with is an `eps_violation`; it raises
P for everything that follows (also when eps is fixed).

The result is "the f32-grade engine's chain unless a pair error exceeded P on an unflagged, unaudited draw".  The F32_SPLIT
engine's logits do not depend on the batch a sample is evaluated in, so with it as `exact` the certified chain IS that
engine's chain, id for id (tests/test_gpu_strict.py, profiles/r05_certified_soak.txt).  No reference counterpart — the
reference has one precision.
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

_GAP_GRID = (0.25, 0.5, 1.0, 2.0, 4.0)      # stats["rerun_share_vs_eps"]: multiples of the eps in use


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


class CertifiedSampler:
    """fast: a reduced-precision Engine (f16 with the f32-grade head recommended: its logit error is 8x below bf16's, so 8x
    fewer close calls); exact: an f32-grade Engine of the same checkpoint (precision 'f32_split' or 'f32')."""

    def __init__(self, fast: Engine, exact: Engine, eps: Optional[float] = None, *, k_sigma: float = 6.0,
                 max_factor: float = 1.05, eps_floor: float = 1e-5, audit_rate: float = 0.02, verify_batch: int = 32,
                 n_boot: int = 8, boot_factor: float = 1.5, audit_seed: int = 0):
        if fast.device != exact.device:
            raise ValueError("both engines must live on the same GPU")
        if eps is not None and not eps > 0:
            raise ValueError("eps must be positive")
        if not k_sigma > 0 or not max_factor >= 1.0 or not boot_factor >= 1.0:
            raise ValueError("k_sigma must be positive, max_factor and boot_factor >= 1")
        if not 0.0 <= audit_rate <= 1.0:
            raise ValueError("audit_rate must be in [0, 1]")
        if verify_batch < 1:
            raise ValueError("verify_batch must be >= 1")
        self.fast, self.exact = fast, exact
        self.eps = None if eps is None else float(eps)          # None: from the error distribution (module docstring)
        self.k_sigma, self.max_factor, self.eps_floor = float(k_sigma), float(max_factor), float(eps_floor)
        self.audit_rate, self.verify_batch = float(audit_rate), int(verify_batch)
        self.n_boot, self.boot_factor = int(n_boot), float(boot_factor)
        self._audit_rng = np.random.default_rng(audit_seed)
        # running error estimate of this engine pair (all calls): per-item r.m.s. and maxima
        self.sigma_d_seen = 0.0      # largest per-ROW r.m.s. of the pair error d (what k_sigma multiplies)
        self.sigma_d_item_seen = 0.0 # largest per-item (sample-update) r.m.s. of d, for the reports
        self.sigma_e_seen = 0.0      # largest per-item r.m.s. of the logit error e
        self.max_d_seen = 0.0
        self.err_seen = 0.0          # largest |e| (r04's statistic, kept for the reports)
        self.n_seen = 0              # verified items the estimate rests on
        self.pair_raise = 0.0        # pair bound forced by violations (also with a fixed eps)
        self.stats: dict = {}

    # ---- eps ------------------------------------------------------------------------------------------------------------
    def pair_bound(self) -> float:
        """P: the bound on the error of a logit difference the next update is certified with (= 2 eps)."""
        if self.eps is not None:
            return max(2.0 * self.eps, self.pair_raise)
        p = max(self.k_sigma * self.sigma_d_seen, self.max_factor * self.max_d_seen)
        if self.n_seen < self.n_boot:
            p *= self.boot_factor
        return max(p, self.pair_raise, 2.0 * self.eps_floor)

    def _eps_now(self) -> float:
        return 0.5 * self.pair_bound()

    def _observe_items(self, st: np.ndarray, V: int) -> None:
        """st (n, 6): per item max |e|, sum e^2, max |d|, sum d^2, masked rows, largest per-row sum d^2."""
        for me, se, md, sd, rows, sd_row in st:
            if rows <= 0:
                continue
            self.n_seen += 1
            self.err_seen = max(self.err_seen, float(me))
            self.max_d_seen = max(self.max_d_seen, float(md))
            self.sigma_e_seen = max(self.sigma_e_seen, math.sqrt(float(se) / (rows * (V - 1))))
            self.sigma_d_item_seen = max(self.sigma_d_item_seen, math.sqrt(float(sd) / (rows * (V - 2))))
            self.sigma_d_seen = max(self.sigma_d_seen, math.sqrt(float(sd_row) / (V - 2)))

    def _item_stats(self, lg_fast: torch.Tensor, lg_exact: torch.Tensor, x_in: torch.Tensor) -> torch.Tensor:
        """(n, 6) on the device: max |e|, sum e^2, max |d|, sum d^2, masked rows, largest per-row sum d^2 — per sample over its
        masked rows."""
        st = self.exact.logit_error_stats(lg_fast, lg_exact, x_in)
        rows = (x_in == STRUCTURE_MASK_TOKEN).sum(1).to(torch.float32)
        return torch.stack([st[..., 0].amax(1), st[..., 1].sum(1), st[..., 2].amax(1), st[..., 3].sum(1), rows, st[..., 3].amax(1)], 1)

    # ---- the loop ---------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def ddpm_sample(self, sequence_tokens: torch.Tensor, schedule: DDPMSchedule, *, seed: int, sample_offset: int = 0,
                    input_prior: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same arguments and result layout as Engine.ddpm_sample.  self.stats describes the call.
        The batch may hold MORE samples than fast.max_batch (the CLI's batch loop, sample_esmdiff.py:181-216, as one stream): every
        fast forward takes the fast.max_batch unfinished samples with the lowest indices, so a sample that was rolled back
        rides along with later ones instead of leaving the GPU with a tail of one-sample forwards.  Ids do not depend on the
        lane width (Philox is keyed by the global sample index)."""
        import time
        fast, exact = self.fast, self.exact
        B, L = sequence_tokens.shape
        W = min(B, fast.max_batch)                                 # lane width: samples per fast forward
        t_call = time.perf_counter()
        dev = fast.device
        MASK = STRUCTURE_MASK_TOKEN
        seq = sequence_tokens.to(device=dev, dtype=torch.int64).contiguous()
        if input_prior is None:
            x = torch.full((B, L), MASK, dtype=torch.int64, device=dev)
        else:
            if tuple(input_prior.shape) != (B, L):
                raise ValueError(f"Invalid input_prior shape: {tuple(input_prior.shape)} v.s. (seq) {(B, L)}")
            x = input_prior.to(device=dev, dtype=torch.int64).contiguous().clone()
        if hasattr(fast, "_check_ids"):
            fast._check_ids(seq, x)                                # once: every later forward sees these ids or the kernels' own
        T = schedule.num_steps
        V = exact.cfg.n_structure_heads
        tf_fast = fast.conditioning_rows(schedule.t_freq)
        tf_exact = exact.conditioning_rows(schedule.t_freq)
        tf_fast_d = None if tf_fast is None else tf_fast.to(device=dev, dtype=torch.float32).contiguous()
        tf_exact_d = None if tf_exact is None else tf_exact.to(device=dev, dtype=torch.float32).contiguous()
        mc_t = np.array([float(schedule.mc_t[i]) for i in range(T)] + [0.0], dtype=np.float32)
        mc_s = np.array([float(schedule.mc_s[i]) for i in range(T)] + [0.0], dtype=np.float32)

        ld = fast.ld_logits
        lbuf = [torch.empty(W, L, ld, dtype=torch.float32, device=dev) for _ in range(2)]     # fast logits, two updates in flight
        cap = max(1, min(exact.max_batch, max(2 * self.verify_batch, 64)))   # a full pool is verified at once, whatever verify_batch
        pools = [{"before": torch.empty(cap, L, dtype=torch.int64, device=dev), "after": torch.empty(cap, L, dtype=torch.int64, device=dev),
                  "lg": torch.empty(cap, L, ld, dtype=torch.float32, device=dev), "items": []} for _ in range(2)]
        fill = 0                                                   # index of the pool that is filling

        step = np.zeros(B, dtype=np.int64)                         # next update of each sample (T = noise removal, T + 1 = done)
        epoch = np.zeros(B, dtype=np.int64)
        mask_known = np.zeros(B, dtype=bool)                       # is has_mask[s] valid for the sample's current state?
        has_mask = np.ones(B, dtype=bool)
        updates: deque = deque()                                   # fast updates whose flags are still on their way
        verifies: deque = deque()
        st = {"samples": B, "updates": T + 1, "fast_launches": 0, "verify_launches": 0, "verify_batch_sizes": [],
              "flagged": 0, "audit_checked": 0, "audit_mismatches": 0, "audit_eps_violations": 0, "audit_max_logit_err": 0.0,
              "audit_max_pair_err": 0.0, "corrections": 0, "eps_violations": 0, "rollback_updates_discarded": 0,
              "sample_forwards_fast": 0, "sample_forwards_exact": 0, "max_logit_err_observed": 0.0, "max_pair_err_observed": 0.0,
              "flagged_per_update": [0] * (T + 1), "eps_used": []}
        gap_log: List[float] = []                                  # min gap of every live sample-update / the eps it ran with
        timers: list = []                                          # (lane, start event, end event) of every forward + draw

        def settle_final(s: int) -> None:
            # a sample without a MASK is carried through every remaining update and through the noise removal unchanged
            # (model.py:530-532, 575-579, 606-607): nothing left to run for it
            if mask_known[s] and not has_mask[s]:
                step[s] = T + 1

        # -- first update of a one-protein batch: identical inputs for every sample -> ONE f32-grade forward, nothing to certify
        shared0 = input_prior is None and B > 1 and T > 0 and bool((seq == seq[:1]).all())
        if input_prior is not None or T == 0:
            has_mask[:], mask_known[:] = (x == MASK).any(1).cpu().numpy(), True
            for s in range(B):
                settle_final(s)
        if self.eps is None and self.n_seen == 0:
            n = 1 if shared0 else min(2, B)                        # start the error estimate (first call of this engine pair)
            t0 = None if tf_exact_d is None else tf_exact_d[0]
            lg_e = exact.forward_logits(x[:n], seq[:n], t0)
            lg_f = fast.forward_logits(x[:n], seq[:n], None if tf_fast_d is None else tf_fast_d[0])
            self._observe_items(self._item_stats(lg_f, lg_e, x[:n]).cpu().numpy(), V)
            st["sample_forwards_fast"] += n
            st["sample_forwards_exact"] += n
        if shared0:
            lg1 = exact.forward_logits(x[:1], seq[:1], None if tf_exact_d is None else tf_exact_d[0])
            lbuf[0][..., :V] = lg1
            for c0 in range(0, B, W):
                n = min(W, B - c0)
                exact.ddpm_step(x[c0:c0 + n], lbuf[0][:n, :, :V], float(mc_t[0]), float(mc_s[0]), seed=seed,
                                sample_offset=sample_offset + c0, step=0)
            step[:] = 1
            st["sample_forwards_exact"] += 1
            if T == 1:
                has_mask[:], mask_known[:] = (x == MASK).any(1).cpu().numpy(), True
                for s in range(B):
                    settle_final(s)

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

        def launch_fast(active: np.ndarray, which: int) -> None:
            n = len(active)
            full = n == B
            steps = step[active].copy()
            # every upload first: nothing below waits for the host
            par = up(fast.sample_step_params_host(sample_offset + active, mc_t[steps], mc_s[steps], steps, (steps == T).astype(np.int32)))
            mixed = not (steps == steps[0]).all()
            steps_d = up(steps) if (mixed and tf_fast_d is not None) else None
            if full:
                xa, sa = x, seq
            else:
                idx_d = up(active)
                xa, sa = x[idx_d], seq[idx_d]
            prev = xa.clone()
            if tf_fast_d is None:
                tf = None
            elif not mixed:
                tf = tf_fast_d[int(steps[0])]
            else:
                tf = tf_fast_d[steps_d]
            e0 = tick()
            lg = fast.forward_logits(xa, sa, tf, out=lbuf[which][:n], check_ids=False)    # ids checked once below
            eps_i = self._eps_now()
            flags = torch.zeros(n, dtype=torch.int32, device=dev)
            gaps = torch.full((n,), float("inf"), dtype=torch.float32, device=dev)
            fast.ddpm_step_rows(xa, lg, par, seed=seed, eps=eps_i, flags=flags, gaps=gaps)
            timers.append(("fast", e0, tick()))
            if not full:
                x[idx_d] = xa
            after = xa.clone() if full else xa
            back = _Async(torch.stack([flags.to(torch.float32), gaps, (after == MASK).any(1).to(torch.float32)]))
            updates.append({"active": active, "steps": steps, "epochs": epoch[active].copy(), "prev": prev, "after": after,
                            "lg": lbuf[which], "back": back, "eps": eps_i})
            step[active] += 1
            mask_known[active] = False
            st["fast_launches"] += 1
            st["sample_forwards_fast"] += n
            st["eps_used"].append(eps_i)

        def process_update(rec) -> None:
            nonlocal fill
            fl, gp, hm = rec["back"].get()
            live = epoch[rec["active"]] == rec["epochs"]
            pick, kinds = [], []
            audits_wanted = 2 if (self.eps is None and self.n_seen < self.n_boot) else 0
            for j in np.nonzero(live)[0]:
                s, k = int(rec["active"][j]), int(rec["steps"][j])
                has_mask[s], mask_known[s] = bool(hm[j] > 0), True
                settle_final(s)
                if np.isfinite(gp[j]):
                    gap_log.append(float(gp[j]) / (2.0 * rec["eps"]))
                if fl[j] > 0:
                    pick.append(j); kinds.append("flag")
                    st["flagged"] += 1
                    st["flagged_per_update"][k] += 1
                elif np.isfinite(gp[j]) and (self._audit_rng.random() < self.audit_rate or audits_wanted > 0):
                    pick.append(j); kinds.append("audit")          # (a sample-update that drew nothing has nothing to audit)
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
                                          "kind": kind, "eps": rec["eps"]})

        def launch_verify() -> None:
            nonlocal fill
            pool = pools[fill]
            items = pool["items"]
            n = len(items)
            if n == 0:
                return
            while verifies:                                       # settle the earlier batch first: its roll-backs make entries stale
                process_verify(verifies.popleft())
            items = pool["items"]
            ss = np.array([it["s"] for it in items], dtype=np.int64)
            ks = np.array([it["k"] for it in items], dtype=np.int64)
            par = up(exact.sample_step_params_host(sample_offset + ss, mc_t[ks], mc_s[ks], ks, (ks == T).astype(np.int32)))
            ss_d, ks_d = up(ss), up(ks)
            xs = pool["before"][:n].clone()
            sq = seq[ss_d]
            tf = None if tf_exact_d is None else tf_exact_d[ks_d]
            if tf is not None and n == 1:
                tf = tf[0]
            e0 = tick()
            lg2 = exact.forward_logits(xs, sq, tf, check_ids=False)
            stats = self._item_stats(pool["lg"][:n], lg2, pool["before"][:n])
            exact.ddpm_step_rows(xs, lg2, par, seed=seed)
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
            self._observe_items(res[:, :6], V)
            for j, it in enumerate(rec["items"]):
                me, md, neq, hm = float(res[j, 0]), float(res[j, 2]), res[j, 6] > 0, res[j, 7] > 0
                st["max_logit_err_observed"] = max(st["max_logit_err_observed"], me)
                st["max_pair_err_observed"] = max(st["max_pair_err_observed"], md)
                audit = it["kind"] == "audit"
                if audit:
                    st["audit_checked"] += 1
                    st["audit_max_logit_err"] = max(st["audit_max_logit_err"], me)
                    st["audit_max_pair_err"] = max(st["audit_max_pair_err"], md)
                if md > 2.0 * it["eps"]:                         # the pair error left the bound this update was certified with
                    st["eps_violations"] += 1
                    st["audit_eps_violations"] += int(audit)
                    self.pair_raise = max(self.pair_raise, self.max_factor * md)
                s = it["s"]
                if epoch[s] != it["epoch"] or not neq:
                    continue                                      # stale (the sample was rolled back meanwhile) or confirmed
                st["audit_mismatches" if audit else "corrections"] += 1
                if audit:                                         # a miss of the certificate: widen the bound for what follows
                    self.pair_raise = max(self.pair_raise, self.max_factor * max(md, 2.0 * it["eps"]) * 1.5)
                st["rollback_updates_discarded"] += int(step[s] - (it["k"] + 1))
                x[s] = rec["xs"][j]
                step[s] = it["k"] + 1
                epoch[s] += 1
                has_mask[s], mask_known[s] = bool(hm), True
                settle_final(s)

        which = 0
        t_tail = None
        while True:
            active = np.nonzero((step < T) | ((step == T) & mask_known & has_mask))[0][:W]
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
            while verifies:                                       # (launched in an earlier iteration, or nothing else to do)
                process_verify(verifies.popleft())
            queued = len(pools[fill]["items"])
            if queued >= self.verify_batch or (queued and not len(active)):
                launch_verify()
            if not len(active) and not updates and not verifies and not pools[fill]["items"]:
                if (step == T + 1).all():
                    break
                if not ((step < T) | ((step == T) & mask_known & has_mask)).any():
                    raise RuntimeError("certified sampler stalled: unfinished samples without pending work")   # (unreachable)

        if fast.device.type == "cuda":
            torch.cuda.synchronize(fast.device)
        st["seconds"] = round(time.perf_counter() - t_call, 4)
        st["tail_seconds"] = 0.0 if t_tail is None else round(time.perf_counter() - t_tail, 4)     # after the last full-lane forward
        st["lane_width"] = W
        if timers and timers[0][1] is not None:                   # device time inside the two lanes (HIP events on the stream)
            for lane in ("fast", "verify"):
                st[f"gpu_seconds_{lane}"] = round(sum(a.elapsed_time(b) for ln_, a, b in timers if ln_ == lane) * 1e-3, 4)
        gl = np.array(gap_log) if gap_log else np.zeros(0)
        n_upd = max(1, len(gl))
        used = st.pop("eps_used")
        st.update({
            "eps": self.eps if self.eps is not None else "auto", "k_sigma": self.k_sigma, "max_factor": self.max_factor,
            "audit_rate": self.audit_rate, "verify_batch": self.verify_batch,
            "eps_min_used": min(used) if used else None, "eps_max_used": max(used) if used else None,
            "sigma_pair_err": self.sigma_d_seen, "sigma_pair_err_per_item": self.sigma_d_item_seen, "sigma_logit_err": self.sigma_e_seen, "max_pair_err_all_calls": self.max_d_seen,
            "max_logit_err_all_calls": self.err_seen, "items_seen_all_calls": self.n_seen, "first_update_shared": shared0,
            # share of sample-updates whose smallest gap is within 2 * (m * eps): what a bound of m x the eps in use would re-run
            "rerun_share_vs_eps": {str(m): round(float((gl <= m).sum()) / n_upd, 5) for m in _GAP_GRID},
            "rerun_share": round(st["flagged"] / n_upd, 5),
        })
        self.stats = st
        return x
