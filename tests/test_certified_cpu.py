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

    def conditioning_rows(self, t_freq):
        return t_freq

    def _clean(self, x, seq, tf):
        x = x.numpy()
        emb = self.table[np.minimum(x, 4101)] + 0.3 * self.table[seq.numpy() % 64]
        ctx = emb.mean(axis=1, keepdims=True)                      # couples the rows of a sample, like attention
        h = np.tanh(emb + ctx + (0.0 if tf is None else float(tf[:8].sum()) * 0.05))
        return (h @ self.proj * self.scale / 8.0).astype(np.float32)

    def forward_logits(self, x, seq, tf, out=None):
        lg = self._clean(x, seq, tf)
        if self.noise:
            self.calls += 1
            rng = np.random.default_rng(self.seed * 7919 + self.calls)
            lg = lg + rng.uniform(-self.noise, self.noise, lg.shape).astype(np.float32)
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

    def ddpm_step_margin(self, x, lg, mc_t, mc_s, *, final, seed, sample_offset, step, margin, flags):
        """esmdiff_ddpm_step_margin restated: ids from the oracle's float32 step, the runner-up test in float64 on the same
        Philox uniforms (made slightly wider than the kernel's, 1e-5 relative, so float32 / float64 rounding cannot un-flag)."""
        xin = x.numpy().copy()
        B, L = xin.shape
        z = lg.numpy().astype(np.float64).copy()
        z[..., MASK] -= 1e6
        lp = z - np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1, keepdims=True)) - z.max(-1, keepdims=True)
        for b in range(B):
            for l in range(L):
                if xin[b, l] != MASK:
                    continue
                if final:
                    val = lp[b, l]
                    top = np.partition(val, -2)[-2:]
                    close = top[1] - top[0] <= margin + 1e-5
                else:
                    u = c_oracle.philox_uniforms(seed, sample_offset + b, step, l, V).astype(np.float64)
                    q = np.exp(lp[b, l]) * (mc_t - mc_s)
                    q[MASK] = mc_s
                    val = q / (1e-10 - np.log(u + 1e-10))
                    top = np.partition(val, -2)[-2:]
                    close = top[1] <= top[0] * margin * (1 + 1e-5)
                if close:
                    flags[b] = 1
        return self.ddpm_step(x, lg, mc_t, mc_s, final=final, seed=seed, sample_offset=sample_offset, step=step)

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
    for trial in range(3):                                           # another perturbation every time
        fast = _Net(noise=noise, seed=10 + trial)
        # eps=None: the bound is 2 x the largest error seen; the noise is uniform in +-0.05, so 2 x max is >= 0.05 after the probes
        cs = CertifiedSampler(fast, _Net(), eps=eps_arg)
        got = cs.ddpm_sample(seq, sch, seed=11)
        st = cs.stats
        assert torch.equal(got, want), (trial, st)
        assert st["eps_violations"] == 0 and st["max_logit_err_observed"] <= noise * 1.0001
        assert st["first_update_shared"] == same_protein
        assert 0 < st["sample_forwards_exact"] < st["sample_forwards_fast"] + B     # some close calls, not everything re-run
        if eps_arg is None:
            assert st["eps_min_used"] >= 0.05 * 0.9 and st["eps_max_used"] <= 2 * noise * 1.0001


def test_certified_with_prior_and_final_pass():
    """input_prior (inpainting: most rows given, a window masked): no step-0 sharing, carried rows untouched; one update only, so
    MASKs survive into the noise-removal pass and its margin rule (difference of log-probabilities) is exercised."""
    B, L, T = 4, 10, 1
    sch = ddpm_schedule(T, freq_dim=256)
    seq = _seqs(B, L, True)
    g = torch.Generator().manual_seed(9)
    prior = torch.randint(0, 4096, (B, L), generator=g)
    prior[:, 2:9] = MASK
    want = _Net().chain(seq, sch, seed=4, prior=prior)
    assert int((want == MASK).sum()) == 0
    cs = CertifiedSampler(_Net(noise=0.05, seed=3), _Net(), eps=0.05)
    got = cs.ddpm_sample(seq, sch, seed=4, input_prior=prior)
    assert torch.equal(got, want) and not cs.stats["first_update_shared"]
    assert torch.equal(got[:, :2], prior[:, :2]) and torch.equal(got[:, 9:], prior[:, 9:])
    assert len(cs.stats["rerun_per_update"]) == T + 1
    with pytest.raises(ValueError, match="input_prior shape"):
        cs.ddpm_sample(seq, sch, seed=4, input_prior=prior[:, :5])


def test_certified_argument_checks():
    with pytest.raises(ValueError, match="eps"):
        CertifiedSampler(_Net(), _Net(), eps=0.0)
    with pytest.raises(ValueError, match="safety"):
        CertifiedSampler(_Net(), _Net(), safety=0.5)
