"""CPU: the "gibbs" oracle.  The canonical-order C restatement (what the HIP kernel is compared with bit for bit) against
the plain-torch statement of the same semantics (sort-based nucleus, log_softmax entropy, top-k by entropy).
PARITY UNPINNED vs the reference: esm's iterative_sampling_raw is not available here (oracle/gibbs_ref.py header)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import gibbs_ref as G


def _case(B, L, seed, scale):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, L, 4104, generator=g) * scale
    u = torch.rand(B, L, 4096, generator=g)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    x = torch.full((B, L), 4096, dtype=torch.int64)
    x[:, 0], x[:, -1] = 4098, 4097
    x[0, 3] = 17
    return logits, u, seq, x


def test_schedule_cosine_unmasking():
    from esmdiff_amd.gibbs import unmask_schedule
    for total, steps in ((256, 16), (256, 50), (64, 50), (5, 16), (1, 3)):
        s = G.unmask_schedule(total, steps)
        assert s == unmask_schedule(total, steps)
        assert len(s) == min(steps, total) and sum(s) == total and all(k >= 0 for k in s)
    assert G.unmask_schedule(256, 4) == [20, 55, 83, 98]   # 256 - int(cos(pi/8)*256 + .1) = 20, ...
    # the other registered schedules: every position unmasked exactly once, linear in equal shares
    from esmdiff_amd.gibbs import NOISE_SCHEDULES
    assert unmask_schedule(256, 4, "linear") == [64, 64, 64, 64]
    for name in NOISE_SCHEDULES:
        for total, steps in ((256, 16), (64, 50), (7, 3)):
            s = unmask_schedule(total, steps, name)
            assert len(s) == min(steps, total) and sum(s) == total and all(k >= 0 for k in s), (name, s)
    with pytest.raises(ValueError, match="unknown schedule"):
        unmask_schedule(10, 3, "sigmoid")


@pytest.mark.parametrize("vocab", [4096, 4101])
def test_c_oracle_matches_torch_semantics(vocab):
    """Sort-based (torch) and threshold-search (C, what the HIP kernel mirrors) forms of the same rule, for the stock
    4096-way head and the ESMDiff 4101-way head (entropy / nucleus over the whole row, specials masked after top-p)."""
    for (B, L, seed, scale, temp, top_p) in ((2, 9, 0, 2.0, 1.4, 0.9), (3, 12, 1, 4.0, 0.7, 0.5), (2, 8, 2, 1.0, 1.0, 1.0)):
        logits, u, seq, x = _case(B, L, seed, scale)
        logits[:, 1::3, 4097] += 3.0 * scale          # a special id inside the nucleus of some rows (only seen when vocab = 4101)
        n_un = torch.tensor([3, 2, 4][:B], dtype=torch.int32)
        xr, ent_r, smp_r = G.gibbs_step_ref(x, seq, logits, temp, top_p, n_un, u, vocab=vocab)
        xc, ent_c, smp_c = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), temp, top_p, n_un.numpy(),
                                               u=u.numpy(), return_aux=True, vocab=vocab)
        masked = (x == 4096).numpy()
        np.testing.assert_allclose(ent_c[masked], ent_r.numpy()[masked], rtol=0, atol=3e-5)
        # draws agree everywhere (a disagreement needs a nucleus-boundary or argmax near-tie: none in these cases)
        assert np.array_equal(smp_c[masked], smp_r.numpy()[masked])
        assert np.array_equal(xc, xr.numpy())
        # structure of the update: exactly n_unmask positions changed, never BOS/EOS, known tokens kept
        changed = (xc != x.numpy())
        assert changed.sum(1).tolist() == n_un.tolist()
        assert not changed[:, 0].any() and not changed[:, -1].any() and xc[0, 3] == 17
        assert xc[changed].max() < 4096


@pytest.mark.parametrize("vocab", [4096, 4101])
def test_temperature_zero_is_argmax_of_the_filtered_logits(vocab):
    """temperature = 0 [ESM-RECALL: esm's sample_logits takes the arg-max then]: the draw is the largest kept valid logit — no
    noise enters (any uniforms, any seed: same ids), the nucleus cannot change it (the maximum is always kept) unless the
    maximum is a special id, in which case the best valid id of the nucleus / of the row is taken; entropy ordering unchanged."""
    for (B, L, seed, scale, top_p) in ((2, 9, 0, 2.0, 0.9), (3, 12, 1, 4.0, 0.3), (2, 8, 2, 1.0, 1.0)):
        logits, u, seq, x = _case(B, L, seed, scale)
        logits[:, 1::3, 4097] += 5.0 * scale
        n_un = torch.tensor([3, 2, 4][:B], dtype=torch.int32)
        xr, ent_r, smp_r = G.gibbs_step_ref(x, seq, logits, 0.0, top_p, n_un, u, vocab=vocab)
        xc, ent_c, smp_c = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 0.0, top_p, n_un.numpy(), u=u.numpy(),
                                               return_aux=True, vocab=vocab)
        masked = (x == 4096).numpy()
        assert np.array_equal(smp_c[masked], smp_r.numpy()[masked]) and np.array_equal(xc, xr.numpy())
        assert np.array_equal(smp_c[masked], logits[..., :4096].argmax(-1).numpy()[masked])     # the best valid id, in every case
        x2 = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 0.0, top_p, n_un.numpy(), seed=123, sample_offset=9, step=3,
                                 vocab=vocab)
        assert np.array_equal(x2, xc)                                                            # no noise: the seed does not matter
        x14 = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, top_p, n_un.numpy(), u=u.numpy(), vocab=vocab)
        assert ((x14 != x.numpy()) == (xc != x.numpy())).all()                                   # same positions (entropy order), other ids


def test_full_row_semantics_differ_from_truncated_row():
    """4101-way head: a heavy special id takes nucleus mass and entropy (esm runs top_p_logits and the entropy on the whole
    row and masks invalid ids afterwards), so fewer valid ids survive than on the truncated row; and when the special id
    alone exceeds top_p nothing valid survives and the best valid id is taken."""
    logits, u, seq, x = _case(1, 6, 9, 1.0)
    logits[0, 2, 4100] = 12.0                                   # ~ all of the mass on a special id
    n_un = np.array([4], np.int32)
    _, e96, s96 = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un, u=u.numpy(), return_aux=True, vocab=4096)
    _, e01, s01 = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un, u=u.numpy(), return_aux=True, vocab=4101)
    assert e01[0, 2] < 0.5 < e96[0, 2]                          # the spike dominates the full-row entropy only
    assert s01[0, 2] == int(logits[0, 2, :4096].argmax())       # nothing valid in the nucleus -> best valid id
    assert s01[0, 2] < 4096 and s96[0, 2] < 4096
    _, er, sr = G.gibbs_step_ref(x, seq, logits, 1.4, 0.9, torch.tensor(n_un), u, vocab=4101)
    assert int(sr[0, 2]) == s01[0, 2] and abs(float(er[0, 2]) - e01[0, 2]) < 3e-5


def test_nucleus_keeps_top1_and_respects_mass():
    logits, u, seq, x = _case(1, 6, 5, 6.0)
    # top_p tiny -> only the argmax survives -> the draw is the argmax regardless of the noise
    xc, _, smp = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 1e-6, np.array([4], np.int32),
                                     u=u.numpy(), return_aux=True)
    am = logits[0, :, :4096].argmax(-1).numpy()
    masked = (x[0] == 4096).numpy()
    masked[0] = masked[-1] = False
    assert np.array_equal(smp[0][masked], am[masked])
    assert (xc[0][masked] == am[masked]).all() and xc[0, 3] == 17


def test_philox_step_is_shard_independent():
    logits, _, seq, x = _case(4, 7, 7, 2.0)
    n_un = np.array([2, 2, 2, 2], np.int32)
    full = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un, seed=3, sample_offset=10, step=1)
    part = c_oracle.gibbs_step(x[2:].numpy(), seq[2:].numpy(), logits[2:].numpy(), 1.4, 0.9, n_un[2:], seed=3,
                               sample_offset=12, step=1)
    assert np.array_equal(full[2:], part)


def test_strategy_random_and_invalid_ids_oracle_semantics():
    """GenerationConfig.strategy = "random" and .invalid_ids (r03): the C oracle (what the HIP kernel is bit-compared with)
    against the torch statement.  invalid ids are masked after top-p exactly like the special ids (never drawn, also not as
    the fall-back id); "random" unmasks the k positions with the smallest per-position Philox key — a uniform k-subset —
    and the drawn tokens themselves are unchanged by the strategy."""
    logits, u, seq, x = _case(3, 14, 11, 2.0)
    n_un = np.array([4, 3, 5], np.int32)
    # invalid_ids with explicit uniforms: C oracle == torch semantics, and no invalid id is ever drawn
    masked = (x == 4096).numpy()
    base = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un, u=u.numpy(), return_aux=True, vocab=4101)
    inv = sorted({int(v) for v in base[2][masked].tolist()})[:20] + [4097]         # forbid ids the free draw picks (+ a special: no-op)
    xc, ent_c, smp_c = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un, u=u.numpy(), return_aux=True,
                                           vocab=4101, invalid_ids=inv)
    xr, ent_r, smp_r = G.gibbs_step_ref(x, seq, logits, 1.4, 0.9, torch.tensor(n_un), u, vocab=4101, invalid_ids=[v for v in inv if v < 4096])
    assert np.array_equal(smp_c[masked], smp_r.numpy()[masked]) and np.array_equal(xc, xr.numpy())
    assert not np.isin(smp_c[masked], inv).any() and np.isin(base[2][masked], inv).sum() >= 20
    untouched = ~np.isin(base[2], inv) & masked                                    # rows whose free draw was legal keep it
    assert np.array_equal(smp_c[untouched], base[2][untouched])
    # strategy "random" (Philox): same tokens as "entropy", other positions; the keys reproduce from the Philox column 4352
    e_x, e_ent, e_smp = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un, seed=5, sample_offset=2, step=1,
                                            return_aux=True, vocab=4101)
    r_x, r_key, r_smp = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un, seed=5, sample_offset=2, step=1,
                                            return_aux=True, vocab=4101, strategy="random")
    assert np.array_equal(e_smp[masked], r_smp[masked])
    changed = r_x != x.numpy()
    assert changed.sum(1).tolist() == n_un.tolist() and not changed[:, 0].any() and not changed[:, -1].any()
    assert not np.array_equal(changed, e_x != x.numpy())
    assert ((r_key[masked] >= 0) & (r_key[masked] < 1)).all()
    # the torch statement fed with the same keys picks the same positions
    uu = np.stack([[c_oracle.philox_uniforms(5, 2 + b, 1, l, 4096) for l in range(14)] for b in range(3)])
    t_x, _, _ = G.gibbs_step_ref(x, seq, logits, 1.4, 0.9, torch.tensor(n_un), torch.from_numpy(uu), vocab=4101, pos_key=r_key)
    assert np.array_equal(t_x.numpy(), r_x)
    # over many steps every eligible position is chosen about equally often (uniform subset)
    hits = np.zeros(14)
    for st in range(400):
        rx = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, np.array([3, 3, 3], np.int32), seed=9, step=st,
                                 vocab=4101, strategy="random")
        hits += (rx != x.numpy()).sum(0)
    elig = hits[1:-1][np.arange(1, 13) != 3]                                       # position 3 of sample 0 is known: skip it
    assert hits[0] == hits[-1] == 0 and elig.min() > 0.7 * elig.mean() and elig.max() < 1.3 * elig.mean()


def test_generation_config_options_are_honoured_or_refused():
    """GenerationConfig fields the reference leaves at their defaults (sample_esmdiff.py:116-119) are never silently
    ignored: the schedule, strategy and invalid_ids are used (r03), an unknown strategy / another track is refused before
    any device work."""
    from esmdiff_amd.gibbs import iterative_sampling_raw
    from esmdiff_amd.sdk import ESMProtein, GenerationConfig
    prot = [ESMProtein(sequence="RPDFCLE")]
    with pytest.raises(NotImplementedError):
        iterative_sampling_raw(object(), prot, [GenerationConfig(track="sequence")])
    with pytest.raises(ValueError, match="strategy"):
        iterative_sampling_raw(object(), prot, [GenerationConfig(strategy="greedy")])
    with pytest.raises(NotImplementedError, match="shares"):
        iterative_sampling_raw(object(), prot * 2, [GenerationConfig(), GenerationConfig(strategy="random")])
    with pytest.raises(ValueError, match="unknown schedule"):
        iterative_sampling_raw(type("E", (), {"has_geom": False})(), prot, [GenerationConfig(schedule="sigmoid", num_steps=3)])


def test_condition_on_coordinates_only_false_uses_the_encoder():
    """GenerationConfig.condition_on_coordinates_only = False is read (VERDICT r03 missing item 4): a protein with coordinates
    and no structure tokens gets tokens from the VQ-VAE encoder for residues with finite coordinates (known tokens are not
    sampled), MASK elsewhere; without an encoder the call is refused; True (the default) leaves every position MASK."""
    from esmdiff_amd.gibbs import iterative_sampling_raw
    from esmdiff_amd.sdk import ESMProtein, GenerationConfig

    class Eng:                                      # records what the loop is asked to sample
        has_geom = True
        def set_frames(self, *a):
            pass
        def gibbs_sample(self, seq, x0, table, temperature, top_p, *, seed, sample_offset=0):
            self.x0, self.table = x0.clone(), table.clone()
            return torch.where(x0 == 4096, torch.full_like(x0, 7), x0)

    class Enc:                                      # tokens = residue index + 100 where coordinates are finite
        def encode(self, coords):
            has = torch.isfinite(coords).all(-1).all(-1)
            return torch.where(has, torch.arange(coords.shape[1])[None] + 100, torch.full(has.shape, 4096))

    xyz = torch.randn(7, 3, 3)
    xyz[2:4] = float("inf")                         # the inpainting marker of the reference (sample_esmdiff.py:88-96)
    prot = [ESMProtein(sequence="RPDFCLE", coordinates=xyz)]
    e = Eng()
    out = iterative_sampling_raw(e, prot, [GenerationConfig(num_steps=4, condition_on_coordinates_only=False)], encoder=Enc())
    assert e.x0[0, 1:-1].tolist() == [100, 101, 4096, 4096, 104, 105, 106] and int(e.table.sum()) == 2
    assert out[0].structure_tokens.tolist() == [100, 101, 7, 7, 104, 105, 106]
    with pytest.raises(RuntimeError, match="encoder"):
        iterative_sampling_raw(Eng(), prot, [GenerationConfig(num_steps=4, condition_on_coordinates_only=False)])
    e2 = Eng()
    iterative_sampling_raw(e2, prot, [GenerationConfig(num_steps=4)], encoder=Enc())
    assert int((e2.x0[0, 1:-1] == 4096).sum()) == 7
