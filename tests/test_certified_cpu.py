"""Host logic of esmdiff_amd/certified.py on CPU, with stand-in engines built on the C oracle's sampler.

The property under test is the one the sampler's docstring derives: if every logit of the fast engine is within eps of the exact
engine's, the certified chain equals the exact chain id for id — whatever the perturbation — because a draw whose winner leads
the runner-up by more than exp(2 eps) cannot change, and the other draws are taken from the exact engine.  The stand-in "fast"
engine is the exact one plus bounded noise, with logits flat enough that uncertified chains DO leave the exact chain."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from esmdiff_amd.certified import CertifiedSampler
from esmdiff_amd.schedule import ddpm_schedule
from oracle import c_oracle

V, MASK = 4101, 4096


class _Net:
    """Deterministic stand-in network: logits depend on the token matrix (so a flipped id changes every later update), the row
    and sigma's sinusoid; optional bounded perturbation with its own counter (a different one at every call)."""

    def __init__(self, noise=0.0, seed=0, scale=0.35):
        self.device = torch.device("cpu")
        self.ld_logits = 4104
        self.max_batch = 3                      # smaller than the batch: the re-run loop has to chunk
        self.cfg = SimpleNamespace(n_structure_heads=V)
        self.noise, self.scale, self.calls, self.seed = noise, scale, 0, seed
        self.table = np.random.default_rng(5).standard_normal((4102, 64)).astype(np.float32)
        self.proj = np.random.default_rng(6).standard_normal((64, V)).astype(np.float32)
        self.sample_forwards = 0
        self.inject = None                      # optional hook (net, x, logits) -> logits: a targeted error

    def conditioning_rows(self, t_freq):
        return t_freq

    def _clean(self, x, seq, tf):
        x = x.numpy()
        emb = self.table[np.minimum(x, 4101)] + 0.3 * self.table[seq.numpy() % 64]
        ctx = emb.mean(axis=1, keepdims=True)                      # couples the rows of a sample, like attention
        if tf is None:
            cond = np.float32(0.0)
        else:      # one sinusoid for the batch, or one per sample (esmdiff_forward_logits_sigmas): the same arithmetic per sample
            t2 = tf.numpy().reshape(-1, tf.shape[-1])[:, :8].astype(np.float64).sum(1).astype(np.float32)
            cond = (t2 * np.float32(0.05))[:, None, None]
        h = np.tanh(emb + ctx + cond)
        return (h @ self.proj * self.scale / 8.0).astype(np.float32)

    def forward_logits(self, x, seq, tf, out=None, check_ids=True):
        lg = self._clean(x, seq, tf)
        if self.noise:
            self.calls += 1
            rng = np.random.default_rng(self.seed * 7919 + self.calls)
            lg = lg + rng.uniform(-self.noise, self.noise, lg.shape).astype(np.float32)
        if self.inject is not None:
            lg = self.inject(self, x, lg)
        self.sample_forwards += x.shape[0]
        if out is None:
            out = torch.empty(x.shape[0], x.shape[1], self.ld_logits)
        out[..., :V] = torch.from_numpy(lg)
        return out[..., :V]

    def ddpm_step(self, x, lg, mc_t, mc_s, *, final=False, seed=None, sample_offset=0, step=0, u=None):
        new = c_oracle.ddpm_step(x.numpy(), np.ascontiguousarray(lg.numpy()), mc_t, mc_s, final=final, seed=seed,
                                 sample_offset=sample_offset, step=step)
        x.copy_(torch.from_numpy(new))
        return x

    def sample_step_params_host(self, sample_index, mc_t, mc_s, step, final):
        from esmdiff_amd.engine import Engine
        return Engine.sample_step_params_host(sample_index, mc_t, mc_s, step, final)

    def sample_step_params(self, sample_index, mc_t, mc_s, step, final):
        return torch.from_numpy(self.sample_step_params_host(sample_index, mc_t, mc_s, step, final))

    def ddpm_step_rows(self, x, lg, params, *, seed, eps=None, flags=None, gaps=None):
        """esmdiff_ddpm_step_rows restated: per sample the oracle's float32 step with that sample's own scalars; the runner-up
        test in float64 on the same Philox uniforms (made slightly wider than the kernel's, 1e-5 relative, so float32 / float64
        rounding cannot un-flag)."""
        from esmdiff_amd import _native as N
        rec = params.numpy().view(N.SAMPLE_STEP_DTYPE).reshape(-1)
        B, L = x.shape
        z = lg.numpy().astype(np.float64).copy()
        z[..., MASK] -= 1e6
        mx = z.max(-1, keepdims=True)
        lp = z - np.log(np.exp(z - mx).sum(-1, keepdims=True)) - mx
        for b in range(B):
            r = rec[b]
            fin, mct, mcs = bool(r["final"]), float(r["move_chance_t"]), float(r["move_chance_s"])
            xin = x[b].numpy().copy()
            for l in range(L):
                if xin[l] != MASK or (flags is None and gaps is None):
                    continue
                if fin:
                    val = lp[b, l]
                    top = np.partition(val, -2)[-2:]
                    gap = top[1] - top[0]
                else:
                    u = c_oracle.philox_uniforms(seed, int(r["sample_index"]), int(r["step"]), l, V).astype(np.float64)
                    q = np.exp(lp[b, l]) * (np.float32(mct) - np.float32(mcs))
                    q[MASK] = mcs
                    val = q / (1e-10 - np.log(u + 1e-10))
                    top = np.partition(val, -2)[-2:]
                    gap = np.log(top[1]) - np.log(max(top[0], 1e-300))
                if flags is not None and gap <= 2.0 * eps * (1 + 1e-5) + 1e-7:
                    flags[b] = 1
                if gaps is not None:
                    gaps[b] = min(float(gaps[b]), max(float(gap), 0.0))
            new = c_oracle.ddpm_step(x[b:b + 1].numpy(), np.ascontiguousarray(lg[b:b + 1].numpy()), mct, mcs, final=fin, seed=seed,
                                     sample_offset=int(r["sample_index"]), step=int(r["step"]))
            x[b:b + 1].copy_(torch.from_numpy(new))
        return x

    def logit_error_stats(self, a, b, x, all_columns=False):
        """esmdiff_logit_error_stats restated (8 columns per row, include/esmdiff_hip.h)."""
        za, zb = a[..., :V].numpy().astype(np.float64), b[..., :V].numpy().astype(np.float64)
        e = za - zb
        if not all_columns:
            za, zb, e = (np.delete(t, MASK, axis=-1) for t in (za, zb, e))   # drawable columns; pairs (4095, 4097) do not exist in the kernel
        d = e[..., :-1] - e[..., 1:]
        if not all_columns:
            d = np.delete(d, MASK - 1, axis=-1)

        def H(z):
            lp = z - z.max(-1, keepdims=True)
            lp = lp - np.log(np.exp(lp).sum(-1, keepdims=True))
            return -(np.exp(lp) * lp).sum(-1)
        ha, hb = H(za), H(zb)
        m = (x.numpy() == MASK)[..., None]
        out = np.stack([np.abs(e).max(-1), (e * e).sum(-1), np.abs(d).max(-1), (d * d).sum(-1), e.max(-1) - e.min(-1), ha - hb, hb,
                        np.zeros_like(hb)], -1) * m
        return torch.from_numpy(out.astype(np.float32))

    # ---- gibbs mode --------------------------------------------------------------------------------------------------------
    def gibbs_step_params_host(self, sample_index, step, n_unmask):
        from esmdiff_amd.engine import Engine
        return Engine.gibbs_step_params_host(sample_index, step, n_unmask)

    def gibbs_step_rows(self, x, seq, lg, temperature, top_p, params, *, seed, pair_bound=None, entropy_bound=None, flags=None,
                        gaps=None):
        """esmdiff_gibbs_step_rows restated by oracle/gibbs_margin_ref.py (ids: the C oracle's plain step per prompt)."""
        from oracle import gibbs_margin_ref as M
        rec = params.numpy().view(M.GIBBS_STEP_DTYPE).reshape(-1)
        new, f, g = M.gibbs_step_rows(x.numpy(), seq.numpy(), np.ascontiguousarray(lg.numpy()), temperature, top_p, rec, seed,
                                      R=pair_bound, E=entropy_bound, vocab=V)
        x.copy_(torch.from_numpy(new))
        if flags is not None:
            flags.copy_(torch.from_numpy(f))
        if gaps is not None:
            gaps.copy_(torch.from_numpy(g))
        return x

    def set_frames(self, rot, trans=None, has_frame=None):
        self.frames_seen = getattr(self, "frames_seen", []) + [None if rot is None else tuple(rot.shape)]

    def gibbs_chain(self, seq, x0, table, temperature, top_p, seed):
        x = x0.clone()
        for t in range(table.shape[0]):
            lg = self.forward_logits(x, seq, None)
            new = c_oracle.gibbs_step(x.numpy(), seq.numpy(), np.ascontiguousarray(lg.numpy()), temperature, top_p, table[t].numpy(),
                                      seed=seed, sample_offset=0, step=t, vocab=V)
            x = torch.from_numpy(new)
        return x

    def chain(self, seq, sch, seed, prior=None):
        B, L = seq.shape
        x = torch.full((B, L), MASK, dtype=torch.int64) if prior is None else prior.clone()
        for i in range(sch.num_steps + 1):
            fin = i == sch.num_steps
            lg = self.forward_logits(x, seq, sch.t_freq[i])
            self.ddpm_step(x, lg, 0.0 if fin else float(sch.mc_t[i]), 0.0 if fin else float(sch.mc_s[i]), final=fin, seed=seed,
                           sample_offset=0, step=i)
        return x


def _seqs(B, L, same):
    g = torch.Generator().manual_seed(3)
    s = torch.randint(4, 24, (1 if same else B, L), generator=g)
    return s.expand(B, L).contiguous() if same else s


@pytest.mark.parametrize("same_protein", [True, False])
@pytest.mark.parametrize("eps_arg", [0.05, None])
def test_certified_chain_equals_exact_chain_under_bounded_logit_error(same_protein, eps_arg):
    B, L, T, noise = 7, 9, 6, 0.05
    sch = ddpm_schedule(T, freq_dim=256)
    seq = _seqs(B, L, same_protein)
    exact = _Net()
    want = exact.chain(seq, sch, seed=11)
    fast = _Net(noise=noise, seed=1)
    plain = fast.chain(seq, sch, seed=11)
    assert not torch.equal(plain, want), "the stand-in is too easy: the perturbed chain never leaves the exact one"
    corrections = 0
    for trial in range(3):                                           # another perturbation every time
        fast = _Net(noise=noise, seed=10 + trial)
        # eps = 0.05: the pair bound 2 eps = 0.1 is the largest difference two logits' errors can have (uniform in +-0.05);
        # eps = None: 6 x the r.m.s. pair error (0.05 sqrt(2/3)) = 0.245 > 0.1, from the first probe on
        cs = CertifiedSampler(fast, _Net(), eps=eps_arg, verify_batch=3, audit_rate=0.1, audit_seed=trial, direct_share=1.0)
        got = cs.ddpm_sample(seq, sch, seed=11)
        st = cs.stats
        assert torch.equal(got, want), (trial, st)
        assert st["eps_violations"] == 0 and st["audit_mismatches"] == 0
        assert st["max_logit_err_observed"] <= noise * 1.0001 and st["max_pair_err_observed"] <= 2 * noise * 1.0001
        assert st["first_update_shared"] == same_protein and st["lane_width"] == 3      # 7 samples through a lane of 3 (max_batch)
        assert st["flagged"] > 0 and st["verify_launches"] > 0 and max(st["verify_batch_sizes"]) <= 3     # exact.max_batch = 3
        assert st["flagged"] < st["sample_forwards_fast"]                                # not everything re-run
        corrections += st["corrections"]
        if eps_arg is None:
            assert 0.1 <= st["eps_min_used"] and st["eps_max_used"] <= 1.5 * 0.5 * 6 * 0.05 * 1.01
            assert abs(st["sigma_pair_err"] - 0.05 * (2 / 3) ** 0.5) < 0.004
    # speculation was wrong somewhere (the uncertified chain leaves the exact one) and the roll-back repaired it
    assert corrections > 0


def test_certified_with_prior_and_final_pass():
    """input_prior (inpainting: most rows given, a window masked): no step-0 sharing, carried rows untouched; one update only, so
    MASKs survive into the noise-removal pass and its margin rule (difference of log-probabilities) is exercised; a sample whose
    prior holds no MASK at all is complete before the first forward."""
    B, L, T = 4, 10, 1
    sch = ddpm_schedule(T, freq_dim=256)
    seq = _seqs(B, L, True)
    g = torch.Generator().manual_seed(9)
    prior = torch.randint(0, 4096, (B, L), generator=g)
    prior[:3, 2:9] = MASK
    want = _Net().chain(seq, sch, seed=4, prior=prior)
    assert int((want == MASK).sum()) == 0
    fast = _Net(noise=0.05, seed=3)
    cs = CertifiedSampler(fast, _Net(), eps=0.05, verify_batch=2, direct_share=1.0)
    got = cs.ddpm_sample(seq, sch, seed=4, input_prior=prior)
    assert torch.equal(got, want) and not cs.stats["first_update_shared"]
    assert torch.equal(got[:, :2], prior[:, :2]) and torch.equal(got[:, 9:], prior[:, 9:]) and torch.equal(got[3], prior[3])
    assert len(cs.stats["flagged_per_update"]) == T + 1
    assert cs.stats["sample_forwards_fast"] <= 3 * (T + 1)            # the complete sample never entered a forward
    with pytest.raises(ValueError, match="input_prior shape"):
        cs.ddpm_sample(seq, sch, seed=4, input_prior=prior[:, :5])


def _inject_big_error(target_calls):
    """A fast-engine fault the margin test cannot see: at the given forward calls, +15 on one codebook logit of every masked row —
    that token wins by a wide margin (unflagged), the exact engine draws something else."""
    def hook(net, x, lg):
        net.inject_calls = getattr(net, "inject_calls", 0) + 1
        if net.inject_calls in target_calls:
            lg = lg.copy()
            m = x.numpy() == MASK
            lg[..., 1234] = np.where(m, lg[..., 1234] + 15.0, lg[..., 1234])
        return lg
    return hook


def test_audit_catches_an_error_above_eps_on_an_unflagged_sample():
    """an error > eps on a sample the margin test did NOT flag.  With every unflagged sample-update audited
    the mismatch is found, counted, repaired (the chain equals the exact one again) and eps is raised; with the audit off the
    same fault silently leaves the exact chain — the audit is what catches it."""
    B, L, T = 5, 8, 5
    sch = ddpm_schedule(T, freq_dim=256)
    seq = _seqs(B, L, False)
    want = _Net().chain(seq, sch, seed=21)
    fast = _Net(noise=0.01, seed=2)
    fast.inject = _inject_big_error({2})
    cs = CertifiedSampler(fast, _Net(), eps=0.01, audit_rate=1.0, verify_batch=4, direct_share=1.0)
    got = cs.ddpm_sample(seq, sch, seed=21)
    st = cs.stats
    assert torch.equal(got, want), st
    assert st["audit_mismatches"] >= 1 and st["audit_eps_violations"] >= 1 and st["eps_violations"] >= st["audit_eps_violations"]
    assert st["audit_max_pair_err"] > 14.9 and st["audit_checked"] >= B
    assert st["eps_max_used"] > 7.0 and cs.pair_bound() > 15.0          # the bound was raised for the following updates
    assert st["rollback_updates_discarded"] >= 0
    fast = _Net(noise=0.01, seed=2)
    fast.inject = _inject_big_error({2})
    blind = CertifiedSampler(fast, _Net(), eps=0.01, audit_rate=0.0, verify_batch=4, direct_share=1.0)
    assert not torch.equal(blind.ddpm_sample(seq, sch, seed=21), want)
    assert blind.stats["audit_checked"] == 0 and blind.stats["audit_mismatches"] == 0


def test_rows_step_is_the_plain_step_per_sample():
    """The stand-in's per-sample step (and with it the contract of esmdiff_ddpm_step_rows the GPU test checks against the kernel):
    samples at different updates in one call = each sample alone through the plain step with its own scalars."""
    net = _Net()
    sch = ddpm_schedule(6, freq_dim=256)
    B, L = 4, 7
    seq = _seqs(B, L, False)
    g = torch.Generator().manual_seed(1)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    x[:, :2] = torch.randint(0, 4096, (B, 2), generator=g)
    steps = np.array([0, 3, 6, 2])
    tf = sch.t_freq[torch.from_numpy(steps)]
    lg = net.forward_logits(x, seq, tf)
    mc_t = np.array([float(sch.mc_t[k]) if k < 6 else 0.0 for k in steps], dtype=np.float32)
    mc_s = np.array([float(sch.mc_s[k]) if k < 6 else 0.0 for k in steps], dtype=np.float32)
    par = net.sample_step_params(10 + np.arange(B), mc_t, mc_s, steps, (steps == 6).astype(np.int32))
    got = net.ddpm_step_rows(x.clone(), lg, par, seed=5)
    for b in range(B):
        one = net.forward_logits(x[b:b + 1], seq[b:b + 1], sch.t_freq[int(steps[b])])
        assert torch.equal(one, lg[b:b + 1])                              # one sigma per sample = that sample alone
        want = net.ddpm_step(x[b:b + 1].clone(), one, float(mc_t[b]), float(mc_s[b]), final=bool(steps[b] == 6), seed=5,
                             sample_offset=10 + b, step=int(steps[b]))
        assert torch.equal(got[b:b + 1], want)


def test_certified_argument_checks():
    with pytest.raises(ValueError, match="eps"):
        CertifiedSampler(_Net(), _Net(), eps=0.0)
    with pytest.raises(ValueError, match="k_sigma"):
        CertifiedSampler(_Net(), _Net(), k_sigma=0.0)
    with pytest.raises(ValueError, match="audit_rate"):
        CertifiedSampler(_Net(), _Net(), audit_rate=1.5)
    with pytest.raises(ValueError, match="verify_batch"):
        CertifiedSampler(_Net(), _Net(), verify_batch=0)


# ---- gibbs mode (the CLI's default, /root/reference/slm/sample_esmdiff.py:66-130, :241) ------------------------------------------
def _gibbs_setup(B, L, steps, same, masked=None):
    """Prompts with BOS / EOS and `masked[b]` interior positions to sample (default: all of them): x0, the (T, B) table."""
    from esmdiff_amd.gibbs import unmask_schedule
    seq = _seqs(B, L, same).clone()
    seq[:, 0], seq[:, -1] = 0, 2
    g = torch.Generator().manual_seed(2)
    x0 = torch.randint(0, 4096, (B, L), generator=g)
    x0[:, 0], x0[:, -1] = 4098, 4097
    masked = [L - 2] * B if masked is None else masked
    for b, n in enumerate(masked):
        x0[b, 1:1 + n] = MASK
    T = max(min(steps, n) for n in masked)
    table = torch.zeros(T, B, dtype=torch.int32)
    for b, n in enumerate(masked):
        sch = unmask_schedule(n, steps)
        table[:len(sch), b] = torch.tensor(sch, dtype=torch.int32)
    return seq, x0, table


@pytest.mark.parametrize("same_protein,temperature,top_p", [(False, 1.4, 0.9), (True, 1.4, 0.9), (False, 0.0, 0.7), (False, 0.8, 1.0)])
def test_certified_gibbs_chain_equals_exact_chain_under_bounded_logit_error(same_protein, temperature, top_p):
    """The three decisions of a gibbs step (nucleus membership, temperature race, entropy order) under a bounded logit error:
    the certified chain equals the exact engine's chain, while the uncertified fast chain leaves it.  Bounds from the measured
    error (eps = None): the range of the row's logit error for the pairs, the entropy error for the order."""
    B, L, steps, noise, scale = 7, 14, 5, 0.03, 1.0
    seq, x0, table = _gibbs_setup(B, L, steps, same_protein)
    want = _Net(scale=scale).gibbs_chain(seq, x0, table, temperature, top_p, seed=11)
    assert int((want == MASK).sum()) == 0
    plain = _Net(noise=noise, seed=1, scale=scale).gibbs_chain(seq, x0, table, temperature, top_p, seed=11)
    assert not torch.equal(plain, want), "the stand-in is too easy: the perturbed chain never leaves the exact one"
    fast = _Net(noise=noise, seed=1, scale=scale)
    cs = CertifiedSampler(fast, _Net(scale=scale), verify_batch=3, audit_rate=0.1, direct_share=1.0)
    got = cs.gibbs_sample(seq, x0, table, temperature, top_p, seed=11)
    st = cs.stats
    assert torch.equal(got, want), st
    assert st["mode"] == "gibbs" and st["certificate"] == "k-sigma statistical + audit"
    assert st["eps_violations"] == 0 and st["entropy_violations"] == 0 and st["audit_mismatches"] == 0
    assert st["max_range_err_observed"] <= 2 * noise * 1.0001 and st["max_entropy_err_observed"] <= st["entropy_eps_max_used"]
    assert st["first_update_shared"] == same_protein and st["lane_width"] == 3
    assert st["flagged"] > 0 and st["corrections"] > 0 and max(st["verify_batch_sizes"]) <= 3
    assert sum(st["flag_reasons"].values()) >= st["flagged"]
    if top_p >= 1.0:
        assert st["flag_reasons"]["nucleus"] == 0            # no nucleus: nothing to be unsure about


def test_certified_gibbs_ragged_prompts_frames_and_empty_steps():
    """Prompts with different numbers of masked positions (different step counts, zero rows in the table), a prompt with
    nothing to sample, more prompts than the lane holds, and per-prompt frames handed to every forward of the prompts it runs."""
    B, L, steps = 5, 12, 4
    seq, x0, table = _gibbs_setup(B, L, steps, False, masked=[10, 3, 0, 7, 1])
    scale = 1.0
    want = _Net(scale=scale).gibbs_chain(seq, x0, table, 1.4, 0.9, seed=5)
    assert torch.equal(want[2], x0[2])
    fast, exact = _Net(noise=0.03, seed=4, scale=scale), _Net(scale=scale)
    cs = CertifiedSampler(fast, exact, verify_batch=2, audit_rate=0.2, direct_share=1.0)
    frames = (torch.zeros(B, L, 3, 3), torch.zeros(B, L, 3), torch.ones(B, L, dtype=torch.bool))
    got = cs.gibbs_sample(seq, x0, table, 1.4, 0.9, seed=5, frames=frames)
    assert torch.equal(got, want), cs.stats
    assert fast.frames_seen[-1] is None and exact.frames_seen[-1] is None           # cleared at the end
    assert all(f is None or (f[0] <= 3 and f[1:] == (L, 3, 3)) for f in fast.frames_seen) and len(fast.frames_seen) > 2
    assert cs.stats["sample_forwards_fast"] <= 10 + 3 + 7 + 1 + 2 * B            # no forward for steps that unmask nothing
    with pytest.raises(ValueError, match="n_unmask_table"):
        cs.gibbs_sample(seq, x0, table[:, :3], 1.4, 0.9, seed=5)
    with pytest.raises(ValueError, match="x0 shape"):
        cs.gibbs_sample(seq, x0[:, :5], table, 1.4, 0.9, seed=5)


def test_certified_switches_to_the_exact_engine_when_speculation_cannot_pay():
    """When most sample-updates need the slow lane (here: a fast engine ten times too noisy for the gaps of this stand-in), the
    sampler stops speculating: the rest of the call runs on the exact engine alone, the chain is still the exact one, and far
    fewer fast forwards are spent than the whole job would take."""
    B, L, steps = 8, 14, 6
    seq, x0, table = _gibbs_setup(B, L, steps, False)
    want = _Net(scale=1.0).gibbs_chain(seq, x0, table, 1.4, 0.9, seed=2)
    fast, exact = _Net(noise=0.3, seed=8, scale=1.0), _Net(scale=1.0)
    fast.max_batch = exact.max_batch = 4
    cs = CertifiedSampler(fast, exact, verify_batch=3)                     # direct_share = 0.5, the default
    got = cs.gibbs_sample(seq, x0, table, 1.4, 0.9, seed=2)
    st = cs.stats
    assert torch.equal(got, want), st
    assert st["direct_lane_from_launch"] is not None and st["direct_lane_share_seen"] > 0.5 and st["sample_forwards_direct"] > 0
    assert st["sample_forwards_fast"] <= 30 and st["sample_forwards_direct"] >= 10          # of 48 sample-updates (decided after 16 results)
    # the next call of the same mode starts where this one started out (the CLI runs batch after batch of one protein): no
    # speculation is wasted at its beginning; the lane would leave again as soon as the reported share fell
    assert cs.lane_memory == {"gibbs": True}
    want2 = _Net(scale=1.0).gibbs_chain(seq, x0, table, 1.4, 0.9, seed=3)
    got2 = cs.gibbs_sample(seq, x0, table, 1.4, 0.9, seed=3)
    assert torch.equal(got2, want2) and cs.stats["sample_forwards_fast"] == 0 and cs.stats["sample_forwards_direct"] >= 40
    with pytest.raises(ValueError, match="direct_share"):
        CertifiedSampler(_Net(), _Net(), direct_share=0.0)


def test_gibbs_rows_step_is_the_plain_step_per_prompt():
    """Contract of esmdiff_gibbs_step_rows (the GPU test checks the kernel against the same statement): prompts at different
    steps in one call = each prompt alone through the plain step with its own Philox index, step and count."""
    from oracle import gibbs_margin_ref as M
    net = _Net(scale=1.0)
    B, L = 4, 10
    seq, x0, _ = _gibbs_setup(B, L, 4, False)
    lg = net.forward_logits(x0, seq, None)
    rec = np.zeros(B, dtype=M.GIBBS_STEP_DTYPE)
    rec["sample_index"], rec["step"], rec["n_unmask"] = 20 + np.arange(B), [0, 3, 1, 2], [2, 0, 5, 1]
    got, flags, gaps = M.gibbs_step_rows(x0.numpy(), seq.numpy(), lg.numpy(), 1.4, 0.9, rec, seed=9, R=1e-3, E=1e-5)
    for b in range(B):
        want = c_oracle.gibbs_step(x0[b:b + 1].numpy(), seq[b:b + 1].numpy(), np.ascontiguousarray(lg[b:b + 1].numpy()), 1.4, 0.9,
                                   np.array([rec["n_unmask"][b]], np.int32), seed=9, sample_offset=20 + b, step=int(rec["step"][b]),
                                   vocab=V)
        assert np.array_equal(got[b], want[0])
        assert int((got[b] != x0[b].numpy()).sum()) == rec["n_unmask"][b]
    assert flags[1] == 0 and np.isinf(gaps[1]).all()          # a prompt that unmasks nothing reports nothing
    # the report is monotone in the bounds: wider bounds can only add flags
    _, f_wide, _ = M.gibbs_step_rows(x0.numpy(), seq.numpy(), lg.numpy(), 1.4, 0.9, rec, seed=9, R=0.5, E=0.5)
    assert ((flags & ~f_wide) == 0).all() and (f_wide[[0, 2, 3]] & 4).all()
