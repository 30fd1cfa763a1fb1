"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel, through the C ABI, against the oracle.

Bars: the fused sampler is BIT-EXACT against the C oracle (same canonical float-op order) and reproduces
the reference's golden ids; floating-point kernels (bf16 MFMA GEMM, LayerNorm, attention, whole forward)
are compared with a float32 torch reference of the same op with the tolerance stated at each assert.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MASK, V = 4096, 4101


@pytest.fixture(scope="module")
def tiny():
    from esmdiff_amd.config import TINY
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.weights import random_init_state_dict
    from oracle.esm3_ref import build_from_state_dict

    sd = random_init_state_dict(TINY, seed=1)
    eng = Engine(TINY, sd, max_batch=8, max_len=300)
    net, emb = build_from_state_dict(TINY, sd)
    yield TINY, sd, eng, net, emb
    eng.close()


def _logits(B, L, seed, ld=4104, scale=3.0):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, L, ld, generator=g) * scale
    return z


# ---------------------------------------------------------------------------------------------------
# fused sampler: bit-exact
@pytest.mark.parametrize("B,L,frac_known", [(2, 7, 0.0), (4, 60, 0.3), (3, 258, 0.8)])
def test_sampler_bit_exact_explicit_uniforms(tiny, B, L, frac_known):
    from oracle import c_oracle
    _, _, eng, _, _ = tiny
    z = _logits(B, L, 10 + B)
    g = torch.Generator().manual_seed(99)
    u = torch.rand(B, L, V, generator=g)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    known = torch.rand(B, L, generator=g) < frac_known
    x[known] = torch.randint(0, 4096, (int(known.sum()),), generator=g)
    for mc_t, mc_s in ((0.999, 0.95904), (0.5, 0.46), (0.04, 1e-5)):
        want = c_oracle.ddpm_step(x.numpy(), z.numpy(), mc_t, mc_s, u=u.numpy())
        got = eng.ddpm_step(x.clone().cuda(), z.cuda()[..., :], mc_t, mc_s, u=u.cuda()).cpu().numpy()
        assert np.array_equal(got, want)
        assert np.array_equal(got[known.numpy()], x.numpy()[known.numpy()])  # carry-over
    want = c_oracle.ddpm_step(x.numpy(), z.numpy(), 0.0, 0.0, final=True)
    got = eng.ddpm_step(x.clone().cuda(), z.cuda(), 0.0, 0.0, final=True).cpu().numpy()
    assert np.array_equal(got, want)


def test_sampler_bit_exact_philox(tiny):
    from oracle import c_oracle
    _, _, eng, _, _ = tiny
    B, L = 5, 33
    z = _logits(B, L, 5)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    x[1, 3] = 7
    for step, off in ((0, 0), (3, 100), (24, 2 ** 33 + 5)):
        want = c_oracle.ddpm_step(x.numpy(), z.numpy(), 0.7, 0.66, seed=1234, sample_offset=off, step=step)
        got = eng.ddpm_step(x.clone().cuda(), z.cuda(), 0.7, 0.66, seed=1234, sample_offset=off, step=step)
        assert np.array_equal(got.cpu().numpy(), want)
    # sharding independence: rows drawn as a batch at offset o == the same rows drawn alone at o + b
    full = eng.ddpm_step(x.clone().cuda(), z.cuda(), 0.7, 0.66, seed=9, sample_offset=40, step=2).cpu()
    for b in (0, 4):
        one = eng.ddpm_step(x[b:b + 1].clone().cuda(), z[b:b + 1].contiguous().cuda(), 0.7, 0.66, seed=9,
                            sample_offset=40 + b, step=2).cpu()
        assert torch.equal(one[0], full[b])


def test_sampler_fuzz_vs_c_oracle(tiny):
    """40 random cases against the C oracle, ids bit for bit: logit scales from 0.01 to 300 (near-uniform rows to rows whose
    exp underflows everywhere but at the top), logits quantised to a few levels (hundreds of exact ties: lowest index wins),
    any mask pattern including none and all, move chances down to differences of 1e-6, Philox and explicit uniforms, the final
    pass, padded row strides — and the margin form (esmdiff_ddpm_step_margin) returning the same ids."""
    from oracle import c_oracle
    _, _, eng, _, _ = tiny
    rng = np.random.default_rng(77)
    for case in range(40):
        B, L = int(rng.integers(1, 6)), int(rng.integers(1, 41))
        ld = int(rng.choice([4104, 4352]))
        scale = float(rng.choice([0.01, 1.0, 30.0, 300.0]))
        z = rng.standard_normal((B, L, ld)).astype(np.float32) * scale
        if case % 4 == 1:
            z = np.round(z / max(scale, 1e-3)).astype(np.float32) * np.float32(max(scale, 1e-3))      # few levels: exact ties
        x = np.full((B, L), MASK, dtype=np.int64)
        frac = float(rng.choice([0.0, 0.3, 0.9, 1.0]))
        known = rng.random((B, L)) < frac
        x[known] = rng.integers(0, 4096, int(known.sum()))
        mc_t = float(rng.choice([0.999, 0.5, 0.04, 1e-5 + 2e-6]))
        mc_s = float(max(0.0, mc_t - float(rng.choice([0.04, 1e-6, mc_t]))))
        final = case % 7 == 3
        seed, off, step = int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 1000))
        zt = torch.from_numpy(z).cuda()
        if case % 3 == 0 and not final:
            u = rng.random((B, L, V), dtype=np.float32)
            want = c_oracle.ddpm_step(x, z, mc_t, mc_s, u=u)
            got = eng.ddpm_step(torch.from_numpy(x).cuda(), zt, mc_t, mc_s, u=torch.from_numpy(u).cuda())
        else:
            want = c_oracle.ddpm_step(x, z, mc_t, mc_s, final=final, seed=seed, sample_offset=off, step=step)
            got = eng.ddpm_step(torch.from_numpy(x).cuda(), zt, mc_t, mc_s, final=final, seed=seed, sample_offset=off, step=step)
            flags = torch.zeros(B, dtype=torch.int32, device="cuda")
            gm = eng.ddpm_step_margin(torch.from_numpy(x).cuda(), zt, mc_t, mc_s, final=final, seed=seed, sample_offset=off,
                                      step=step, margin=0.01 if final else 1.01, flags=flags)
            assert torch.equal(gm, got), case
            assert not bool(flags[torch.from_numpy(known.all(1)).cuda()].any()), case    # a sample without a MASK is never flagged
        assert np.array_equal(got.cpu().numpy(), want), (case, B, L, ld, scale, frac, mc_t, mc_s, final)
        assert np.array_equal(want[known], x[known])


def test_sampler_rows_equals_plain_step_per_sample(tiny):
    """esmdiff_ddpm_step_rows (ABI 7): a batch whose samples sit at different updates (own Philox index, move chances, step,
    final flag) = every sample alone through esmdiff_ddpm_step with its scalars, ids bit for bit, and through the C oracle; the
    per-sample flags are esmdiff_ddpm_step_margin's, and min_gap is the smallest log gap of the sample's masked rows (checked
    in float64 from the oracle's uniforms)."""
    from oracle import c_oracle
    _, _, eng, _, _ = tiny
    rng = np.random.default_rng(5)
    for case in range(12):
        B, L = int(rng.integers(1, 9)), int(rng.integers(1, 33))
        ld = int(rng.choice([4104, 4352]))
        z = rng.standard_normal((B, L, ld)).astype(np.float32) * float(rng.choice([0.5, 3.0, 40.0]))
        x = np.full((B, L), MASK, dtype=np.int64)
        known = rng.random((B, L)) < float(rng.choice([0.0, 0.4, 0.95]))
        x[known] = rng.integers(0, 4096, int(known.sum()))
        mct = rng.choice([0.999, 0.6, 0.04], B).astype(np.float32)
        mcs = np.maximum(0, mct - rng.choice([0.04, 1e-4], B)).astype(np.float32)
        steps = rng.integers(0, 50, B)
        fin = (rng.random(B) < 0.3).astype(np.int32)
        idx = rng.integers(0, 2 ** 40, B)
        seed = int(rng.integers(0, 2 ** 31))
        eps = float(rng.choice([1e-3, 0.05]))
        zt = torch.from_numpy(z).cuda()
        par = eng.sample_step_params(idx, mct, mcs, steps, fin)
        flags = torch.zeros(B, dtype=torch.int32, device="cuda")
        gaps = torch.full((B,), float("inf"), device="cuda")
        got = eng.ddpm_step_rows(torch.from_numpy(x).cuda(), zt, par, seed=seed, eps=eps, flags=flags, gaps=gaps)
        plain = eng.ddpm_step_rows(torch.from_numpy(x).cuda(), zt, par, seed=seed)          # no flags requested: same ids
        assert torch.equal(got, plain)
        for b in range(B):
            kw = dict(final=bool(fin[b]), seed=seed, sample_offset=int(idx[b]), step=int(steps[b]))
            one = eng.ddpm_step(torch.from_numpy(x[b:b + 1]).cuda(), zt[b:b + 1], float(mct[b]), float(mcs[b]), **kw)
            assert torch.equal(got[b:b + 1], one), (case, b)
            want = c_oracle.ddpm_step(x[b:b + 1], z[b:b + 1], float(mct[b]), float(mcs[b]), **kw)
            assert np.array_equal(one.cpu().numpy(), want), (case, b)
            f1 = torch.zeros(1, dtype=torch.int32, device="cuda")
            eng.ddpm_step_margin(torch.from_numpy(x[b:b + 1]).cuda(), zt[b:b + 1], float(mct[b]), float(mcs[b]),
                                 margin=2 * eps if fin[b] else float(np.exp(2 * eps)), flags=f1, **kw)
            assert int(f1[0]) == int(flags[b]), (case, b)
            # the gap statistic against a float64 restatement
            zz = z[b, :, :V].astype(np.float64).copy()
            zz[:, MASK] -= 1e6
            lp = zz - (np.log(np.exp(zz - zz.max(-1, keepdims=True)).sum(-1, keepdims=True)) + zz.max(-1, keepdims=True))
            best = np.inf
            for l in range(L):
                if x[b, l] != MASK:
                    continue
                if fin[b]:
                    val = lp[l]
                    top = np.partition(val, -2)[-2:]
                    gap = top[1] - top[0]
                else:
                    u = c_oracle.philox_uniforms(seed, int(idx[b]), int(steps[b]), l, V).astype(np.float64)
                    q = np.exp(lp[l]) * (np.float64(mct[b]) - np.float64(mcs[b]))
                    q[MASK] = mcs[b]
                    val = q / (1e-10 - np.log(u + 1e-10))
                    top = np.partition(val, -2)[-2:]
                    gap = np.log(top[1]) - np.log(max(top[0], 1e-37))
                best = min(best, gap)
            g = float(gaps[b])
            if np.isinf(best):
                assert np.isinf(g) and int(flags[b]) == 0
            else:
                assert abs(g - best) <= 2e-5 * max(1.0, abs(best)) + 2e-6, (case, b, g, best)
    with pytest.raises(RuntimeError, match="margin"):
        eng.ddpm_step_rows(torch.from_numpy(x).cuda(), zt, par, seed=1, eps=-1.0, flags=flags)


@pytest.mark.parametrize("all_columns", [False, True])
def test_logit_error_stats_vs_torch(tiny, all_columns):
    """esmdiff_logit_error_stats (ABI 8: 8 floats per row): per masked row max / sum of squares of the logit error and of the
    neighbouring-pair error, the RANGE max e - min e (the bound on ANY pair's error: the pair that decides a draw is not a
    neighbouring one) and the entropy difference — over the drawable columns (all but MASK; the pairs (4095, 4096) and (4096, 4097)
    do not exist) for the ddpm draw, over every column for the gibbs step; zeros for carried rows — vs float64."""
    _, _, eng, _, _ = tiny
    g = torch.Generator().manual_seed(8)
    n, L = 5, 23
    a = torch.randn(n, L, 4104, generator=g).cuda()
    b = (a.cpu() + 1e-3 * torch.randn(n, L, 4104, generator=g)).cuda()
    if not all_columns:
        b[..., MASK] += 100.0                                    # the MASK column must not count
    # one NON-neighbouring pair with a large error difference on one row: invisible to the pair statistic, seen by the range
    b[1, 2, 100] += 0.05
    b[1, 2, 101] += 0.05
    b[1, 2, 102] += 0.05
    b[1, 2, 3000] -= 0.05
    b[1, 2, 3001] -= 0.05
    b[1, 2, 3002] -= 0.05
    x = torch.full((n, L), MASK, dtype=torch.int64)
    x[:, 3:9] = 7
    x[2] = 5
    got = eng.logit_error_stats(a[..., :V], b, x.cuda(), all_columns).cpu().double()
    assert got.shape == (n, L, 8)
    ad, bd = a[..., :V].cpu().double(), b[..., :V].cpu().double()
    e = ad - bd
    keep = list(range(V)) if all_columns else [v for v in range(V) if v != MASK]
    d = e[..., :-1] - e[..., 1:]
    dk = list(range(V - 1)) if all_columns else [v for v in range(V - 1) if v != MASK and v + 1 != MASK]
    H = lambda z: -(torch.log_softmax(z, -1).exp() * torch.log_softmax(z, -1)).sum(-1)
    ha, hb = H(ad[..., keep]), H(bd[..., keep])
    want = torch.stack([e[..., keep].abs().amax(-1), (e[..., keep] ** 2).sum(-1), d[..., dk].abs().amax(-1), (d[..., dk] ** 2).sum(-1),
                        e[..., keep].amax(-1) - e[..., keep].amin(-1), ha - hb, hb, torch.zeros_like(hb)], -1)
    want = want * (x == MASK)[..., None]
    assert torch.equal(got[x != MASK], torch.zeros_like(got[x != MASK]))
    assert float((got[..., 0] - want[..., 0]).abs().max()) == 0 and float((got[..., 2] - want[..., 2]).abs().max()) == 0
    assert float((got[..., 4] - want[..., 4]).abs().max()) < 1e-7
    assert float(((got[..., 1] - want[..., 1]) / want[..., 1].clamp_min(1e-30)).abs().max()) < 1e-5
    assert float(((got[..., 3] - want[..., 3]) / want[..., 3].clamp_min(1e-30)).abs().max()) < 1e-5
    assert float((got[..., 5] - want[..., 5]).abs().max()) < 3e-6 and float((got[..., 6] - want[..., 6]).abs().max()) < 1e-5
    assert float(got[1, 2, 4]) > 0.099 and float(got[1, 2, 2]) < 0.06         # the injected pair: range 0.1, largest neighbour error 0.05


def _gibbs_case(rng, B, L, scale, frac_known=0.0):
    z = rng.standard_normal((B, L, 4104)).astype(np.float32) * scale
    seq = np.concatenate([[0], rng.integers(4, 24, L - 2), [2]])[None].repeat(B, 0).astype(np.int64)
    x = np.full((B, L), MASK, dtype=np.int64)
    x[:, 0], x[:, -1] = 4098, 4097
    known = rng.random((B, L)) < frac_known
    known[:, 0] = known[:, -1] = False
    x[known] = rng.integers(0, 4096, int(known.sum()))
    return z, seq, x


def test_gibbs_rows_equals_plain_step_per_prompt_and_margin_report(tiny):
    """esmdiff_gibbs_step_rows (ABI 8): a batch whose prompts sit at different steps (own Philox index, step, count) = every prompt
    alone through esmdiff_gibbs_step with its scalars, ids bit for bit, and through the C oracle; with bounds the ids are the
    same and the per-prompt report lies between the float64 restatement's (oracle/gibbs_margin_ref.py) at 2 % tighter and 2 %
    wider bounds — flags are monotone in the bounds, so only decisions within 2 % of a bound may differ."""
    from oracle import c_oracle
    from oracle import gibbs_margin_ref as M
    from esmdiff_amd import _native as N
    _, _, eng, _, _ = tiny
    rng = np.random.default_rng(17)
    n_flag = n_clear = 0
    for case in range(14):
        B, L = int(rng.integers(1, 7)), int(rng.integers(4, 29))
        scale = float(rng.choice([0.3, 0.6, 3.0, 12.0]))
        z, seq, x = _gibbs_case(rng, B, L, scale, float(rng.choice([0.0, 0.4])))
        if case % 4 == 1:
            z[:, ::3, 4099] += 6.0 * scale                   # a heavy special id inside the nucleus of some rows
        temp, top_p = float(rng.choice([0.0, 0.7, 1.4])), float(rng.choice([0.5, 0.9, 1.0]))
        idx, steps, ks = rng.integers(0, 2 ** 40, B), rng.integers(0, 60, B), rng.integers(0, 6, B)
        seed = int(rng.integers(0, 2 ** 31))
        R, E = float(rng.choice([2e-3, 2e-2, 0.2])), float(rng.choice([1e-5, 1e-3, 3e-2]))
        xt, st, zt = torch.from_numpy(x).cuda(), torch.from_numpy(seq).cuda(), torch.from_numpy(z).cuda()
        par = torch.from_numpy(eng.gibbs_step_params_host(idx, steps, ks)).cuda()
        flags = torch.zeros(B, dtype=torch.int32, device="cuda")
        gaps = torch.full((B, 2), float("inf"), device="cuda")
        got = eng.gibbs_step_rows(xt.clone(), st, zt, temp, top_p, par, seed=seed, pair_bound=R, entropy_bound=E, flags=flags, gaps=gaps)
        plain = eng.gibbs_step_rows(xt.clone(), st, zt, temp, top_p, par, seed=seed)
        assert torch.equal(got, plain), case
        for b in range(B):
            one = eng.gibbs_step(xt[b:b + 1].clone(), st[b:b + 1], zt[b:b + 1], temp, top_p, torch.tensor([int(ks[b])], dtype=torch.int32),
                                 seed=seed, sample_offset=int(idx[b]), step=int(steps[b]))
            assert torch.equal(got[b:b + 1], one), (case, b)
        rec = np.zeros(B, dtype=N.GIBBS_STEP_DTYPE)
        rec["sample_index"], rec["step"], rec["n_unmask"] = idx, steps, ks
        want, f_lo, g_ref = M.gibbs_step_rows(x, seq, z, temp, top_p, rec, seed, R=R * 0.98, E=E * 0.98, vocab=4101)
        _, f_hi, _ = M.gibbs_step_rows(x, seq, z, temp, top_p, rec, seed, R=R * 1.02, E=E * 1.02, vocab=4101)
        assert np.array_equal(got.cpu().numpy(), want), case
        f = flags.cpu().numpy()
        assert ((f_lo & ~f) == 0).all() and ((f & ~f_hi) == 0).all(), (case, f.tolist(), f_lo.tolist(), f_hi.tolist(), temp, top_p, R, E)
        gp = gaps.cpu().numpy()
        for b in range(B):
            if ks[b] <= 0 or not (x[b] == MASK).any():
                assert f[b] == 0 and np.isinf(gp[b]).all()
                continue
            for c in range(2):
                if g_ref[b, c] < 1e30:
                    # (temperature 0: the kernel's race values are z - max + 2 against a floor of -1, so its gap saturates at 3)
                    ref_gap = min(float(g_ref[b, c]), 3.0) if (temp == 0.0 and c == 0) else float(g_ref[b, c])
                    assert abs(gp[b, c] - ref_gap) <= 3e-5 * max(1.0, abs(ref_gap)) + 3e-6, (case, b, c, gp[b], g_ref[b])
            n_flag += int(f[b] != 0)
            n_clear += int(f[b] == 0)
    assert n_flag >= 5 and n_clear >= 5, (n_flag, n_clear)       # both outcomes were exercised
    with pytest.raises(RuntimeError, match="pair_bound"):
        lib_flags = torch.zeros(B, dtype=torch.int32, device="cuda")
        eng._chk(eng._lib.esmdiff_gibbs_step_rows(eng._h, xt.data_ptr(), st.data_ptr(), zt.data_ptr(), 4104, 1.0, 0.9, par.data_ptr(), 1, B, L,
                                                  -1.0, 0.0, lib_flags.data_ptr(), None, None))


def test_gibbs_margin_unflagged_prompts_are_invariant_under_bounded_errors(tiny):
    """What the report of esmdiff_gibbs_step_rows certifies: an UNFLAGGED prompt's new ids are those any logits within the
    bounds would have produced.  For random rows, eight error fields whose range is below R — random, and adversarial ones
    (everything above the nucleus cut pushed down and everything below pushed up, and the reverse; the runner-up of the race
    pushed up; low-entropy rows pushed towards high entropy) — and E = the largest entropy change they cause: every unflagged
    prompt draws the same ids from the perturbed logits; flagged prompts exist and some of them do change."""
    _, _, eng, _, _ = tiny
    rng = np.random.default_rng(23)
    tot = {"unflagged": 0, "flagged": 0, "flagged_changed": 0}
    for case in range(6):
        B, L = 6, 24
        scale = float([0.6, 1.0, 3.0, 6.0, 6.0, 10.0][case])        # near-uniform rows (entropies ~1e-3 apart) ... peaked rows (~0.2 apart)
        temp, top_p = (1.4, 0.9) if case != 4 else (0.0, 0.8)
        R = float([0.003, 0.01, 0.01, 0.02, 0.02, 0.05][case])
        z, seq, x = _gibbs_case(rng, B, L, scale)
        ks = rng.integers(1, 6, B)
        idx, steps, seed = np.arange(B) + 50, rng.integers(0, 20, B), 7 + case
        zd = z[..., :V].astype(np.float64)
        p = np.exp(zd - zd.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        order = np.argsort(-zd, -1)
        cum = np.take_along_axis(np.cumsum(np.take_along_axis(p, order, -1), -1), np.argsort(order, -1), -1)   # mass of {z_j >= z_v}
        fields = [rng.uniform(-R / 2, R / 2, zd.shape) * 0.99 for _ in range(4)]
        fields.append(np.where(cum <= top_p, -R / 2, R / 2) * 0.99)         # squeeze the nucleus from both sides
        fields.append(np.where(cum <= top_p, R / 2, -R / 2) * 0.99)
        Hrow = -(p * np.log(np.maximum(p, 1e-300))).sum(-1)
        surpr = -np.log(np.maximum(p, 1e-300)) - Hrow[..., None]
        fields.append(np.sign(surpr) * R / 2 * 0.99)                        # the steepest entropy change a range-R error allows
        fields.append(-np.sign(surpr) * R / 2 * 0.99)

        def H(zz):
            q = np.exp(zz - zz.max(-1, keepdims=True))
            q /= q.sum(-1, keepdims=True)
            return -(q * np.log(np.maximum(q, 1e-300))).sum(-1)
        E = max(float(np.abs(H(zd + f) - H(zd)).max()) for f in fields) * 1.02 + 1e-6
        xt, st = torch.from_numpy(x).cuda(), torch.from_numpy(seq).cuda()
        par = torch.from_numpy(eng.gibbs_step_params_host(idx, steps, ks)).cuda()
        flags = torch.zeros(B, dtype=torch.int32, device="cuda")
        base = eng.gibbs_step_rows(xt.clone(), st, torch.from_numpy(z).cuda(), temp, top_p, par, seed=seed, pair_bound=R, entropy_bound=E,
                                   flags=flags).cpu()
        f = flags.cpu().numpy()
        changed = np.zeros(B, bool)
        for fld in fields:
            zp = z.copy()
            zp[..., :V] = (zd + fld).astype(np.float32)
            other = eng.gibbs_step_rows(xt.clone(), st, torch.from_numpy(zp).cuda(), temp, top_p, par, seed=seed).cpu()
            changed |= (other != base).any(1).numpy()
        assert not (changed & (f == 0)).any(), (case, f.tolist(), changed.tolist())
        tot["unflagged"] += int((f == 0).sum())
        tot["flagged"] += int((f != 0).sum())
        tot["flagged_changed"] += int((changed & (f != 0)).sum())
    assert tot["unflagged"] >= 6 and tot["flagged"] >= 6 and tot["flagged_changed"] >= 1, tot


def test_sampler_full_size_config2(tiny):
    """BASELINE config 2 row count (B*L = 100*258 rows of 4101) — bit-exact on a 24-sample slice, and the
    size-independent properties on all of it: carry-over, ids in range, mask never drawn at mc_s = 0."""
    from oracle import c_oracle
    _, _, eng, _, _ = tiny
    B, L = 100, 258
    g = torch.Generator().manual_seed(3)
    z = (torch.randn(B, L, 4104, generator=g) * 2).cuda()
    x = torch.full((B, L), MASK, dtype=torch.int64)
    x[:, ::5] = 11
    xs = x.cuda()
    out = eng.ddpm_step(xs.clone(), z, 0.5, 0.46, seed=77, step=1).cpu()
    assert torch.equal(out[:, ::5], x[:, ::5])
    assert int(out.min()) >= 0 and int(out.max()) <= 4100
    want = c_oracle.ddpm_step(x[:24].numpy(), z[:24].cpu().numpy(), 0.5, 0.46, seed=77, step=1)
    assert np.array_equal(out[:24].numpy(), want)
    out0 = eng.ddpm_step(xs.clone(), z, 0.04, 0.0, seed=77, step=24).cpu()
    assert int((out0 == MASK).sum()) == 0          # q[MASK] = mc_s = 0 can never win
    fin = eng.ddpm_step(xs.clone(), z, 0.0, 0.0, final=True).cpu()
    zz = z[:, 1, :V].cpu().clone()
    zz[:, MASK] -= 1e6                                # only the MASK column is suppressed (model.py:528)
    assert torch.equal(fin[:, 1], zz.argmax(-1))


def test_sampler_golden_reference_ids(tiny, golden_dir):
    """The kernel reproduces the ids the REFERENCE's own code emitted (tests/golden/g5, g6)."""
    from oracle import sampler_ref as R
    from tests.standin_net import StandinNet, standin_sigma_embedder_state
    _, _, eng, _, _ = tiny
    emb = R.TimestepEmbedderRef(32)
    emb.load_state_dict(standin_sigma_embedder_state(32))
    net = StandinNet(32)
    g6 = np.load(golden_dir / "g6_ddpm_sample.npz")
    for tag in ("T5_noprior", "T25_noprior", "T25_prior", "T5_prior"):
        seq = torch.from_numpy(g6[f"{tag}_seq"])
        T = int(g6[f"{tag}_T"])
        B, L = seq.shape
        x = (torch.from_numpy(g6[f"{tag}_prior"]) if f"{tag}_prior" in g6 else
             torch.full((B, L), MASK, dtype=torch.int64)).cuda()
        s = R.ddpm_schedule_ref(T)
        torch.manual_seed(int(g6[f"{tag}_seed"]))
        with torch.no_grad():
            for i in range(T + 1):
                cond = torch.tile(emb(s["sigma_t"][i] * torch.ones(B))[:, None, :], (1, L, 1))
                raw = net(structure_tokens=x.cpu(), sequence_tokens=seq, auxiliary_embeddings=cond).structure_logits
                if i < T:
                    u = torch.rand(B, L, V)        # the reference's torch.rand_like stream (model.py:27)
                    eng.ddpm_step(x, raw.cuda(), s["mc_t"][i].item(), s["mc_s"][i].item(), u=u.cuda())
                    assert np.array_equal(x.cpu().numpy(), g6[f"{tag}_traj"][i]), (tag, i)
                else:
                    eng.ddpm_step(x, raw.cuda(), 0.0, 0.0, final=True)
        assert np.array_equal(x.cpu().numpy(), g6[f"{tag}_final"]), tag


# ---------------------------------------------------------------------------------------------------
# bf16 MFMA GEMM + epilogues
def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 256, 192), (777, 384, 512), (25800 // 8, 1536, 1536)])
def test_gemm_bf16_store(M, N, K):
    from esmdiff_amd import _native as Nn
    from esmdiff_amd.engine import gemm_bf16
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    W = _bf(torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    out = gemm_bf16(A, W, Nn.EPI_BF16)
    ref = A.float() @ W.float().t()
    # f32 accumulation of exact bf16 products; output rounded to bf16: |err| <= 2^-8 |ref| + accumulation noise
    err = (out.float() - ref).abs()
    assert float((err - ref.abs() * 2 ** -8).max()) < 2e-3, float(err.max())


@pytest.mark.parametrize("M,N,K,epi", [(12900, 4608, 1536, "bf16"), (25800, 1536, 4096, "bf16"), (12900, 8192, 1536, "swiglu")])
def test_gemm256_persistent_multi_tile(M, N, K, epi):
    """The full-size block linears: more tiles than CUs, so every workgroup of the 256x256 kernel walks several tiles
    (cross-tile prefetch, register-direct epilogue, ragged last row tile)."""
    from esmdiff_amd import _native as Nn
    from esmdiff_amd.engine import gemm_bf16
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = _bf(torch.randn(M, K, generator=g, device="cuda"))
    W = _bf(torch.randn(N, K, generator=g, device="cuda") / K ** 0.5)
    for _ in range(2):
        if epi == "bf16":
            out = gemm_bf16(A, W, Nn.EPI_BF16)
            ref = A.float() @ W.float().t()
            err = (out.float() - ref).abs()
            assert float((err - ref.abs() * 2 ** -8).max()) < 4e-3, float(err.max())
        else:
            out = gemm_bf16(A, W, Nn.EPI_SWIGLU_BF16)
            r = (A.float() @ W.float().t()).view(M, N // 64, 2, 32)
            ref = (torch.nn.functional.silu(r[:, :, 0]) * r[:, :, 1]).reshape(M, N // 2)
            err = (out.float() - ref).abs()
            assert float((err - ref.abs() * 2 ** -7).max()) < 1e-2, float(err.max())


@pytest.mark.parametrize("M,N,K", [(240, 1536, 4096), (120, 1536, 4096), (774, 1536, 2048), (240, 512, 4096)])
def test_gemm_small_m_split_k(tiny, monkeypatch, M, N, K):
    """The small-M path as the engine runs it (FFN-down at a handful of short sequences): four-stage ring + K slices
    + the fixed-order reduce kernel, against f32 torch and against the unsplit kernel (same bf16 result up to the f32
    re-association of the slices), twice in a row (the workspace is reused)."""
    from esmdiff_amd import _native as Nn
    from esmdiff_amd.engine import gemm_bf16
    _, _, eng, _, _ = tiny
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = _bf(torch.randn(M, K, generator=g, device="cuda"))
    W = _bf(torch.randn(N, K, generator=g, device="cuda") / K ** 0.5)
    ref = A.float() @ W.float().t()
    plain = gemm_bf16(A, W, Nn.EPI_BF16, alpha=0.866).float()
    for _ in range(2):
        out = eng.gemm(A, W, Nn.EPI_BF16, alpha=0.866).float()
        err = (out - ref * 0.866).abs()
        assert float((err - ref.abs() * 0.866 * 2 ** -8).max()) < 2e-3, float(err.max())
        assert float((out - plain).abs().max()) <= float(ref.abs().max()) * 2 ** -7   # at most one bf16 ulp apart
        assert float((out != plain).float().mean()) < 0.05
    bias = torch.randn(N, generator=g, device="cuda")
    got = eng.gemm(A, W, Nn.EPI_BIAS_GELU_BF16, bias=bias).float()
    want = torch.nn.functional.gelu(ref + bias)
    assert float((got - want).abs().max()) < 3e-2 and float((got - want).abs().mean()) < 2e-3


def test_gemm_asymmetric_identity():
    """A = I picks rows of W^T: catches transposed / permuted C layouts (asymmetric B)."""
    from esmdiff_amd import _native as Nn
    from esmdiff_amd.engine import gemm_bf16
    K = N = 256
    A = _bf(torch.eye(K)).cuda()
    W = _bf(torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125).cuda()
    out = gemm_bf16(A, W, Nn.EPI_BF16)
    assert torch.equal(out.float(), W.float().t())


def test_gemm_epilogues():
    from esmdiff_amd import _native as Nn
    from esmdiff_amd.engine import gemm_bf16
    g = torch.Generator().manual_seed(5)
    M, K = 300, 256
    A = _bf(torch.randn(M, K, generator=g)).cuda()
    # residual: x += (A W^T) * alpha, f32 in place
    W = _bf(torch.randn(512, K, generator=g) / 16).cuda()
    x0 = torch.randn(M, 512, generator=g).cuda()
    x = x0.clone()
    gemm_bf16(A, W, Nn.EPI_RESID_F32, out=x, alpha=0.866)
    ref = x0 + (A.float() @ W.float().t()) * 0.866
    assert float((x - ref).abs().max()) < 2e-4
    # SwiGLU: rows interleaved [gate 32 | up 32]; out = silu(gate) * up
    H = 384
    Wg = _bf(torch.randn(H, K, generator=g) / 16)
    Wu = _bf(torch.randn(H, K, generator=g) / 16)
    Wi = torch.stack([Wg.view(H // 32, 32, K), Wu.view(H // 32, 32, K)], 1).reshape(2 * H, K).contiguous().cuda()
    out = gemm_bf16(A, Wi, Nn.EPI_SWIGLU_BF16)
    gate, up = A.float() @ Wg.float().cuda().t(), A.float() @ Wu.float().cuda().t()
    ref = torch.nn.functional.silu(gate) * up
    assert out.shape == (M, H)
    assert float((out.float() - ref).abs().max()) < 2e-2 and float((out.float() - ref).abs().mean()) < 1e-3
    # bias + GELU (erf)
    bias = torch.randn(512, generator=g).cuda()
    out = gemm_bf16(A, W, Nn.EPI_BIAS_GELU_BF16, bias=bias)
    ref = torch.nn.functional.gelu(A.float() @ W.float().t() + bias)
    assert float((out.float() - ref).abs().max()) < 3e-2 and float((out.float() - ref).abs().mean()) < 2e-3
    # bias, f32 out, ragged n_valid with padded leading dimension
    nv, ld = 389, 392
    o = torch.full((M, ld), -7.0, device="cuda")
    gemm_bf16(A, W, Nn.EPI_BIAS_F32, out=o, bias=bias, n_valid=nv)
    ref = A.float() @ W.float().t() + bias
    assert float((o[:, :nv] - ref[:, :nv]).abs().max()) < 2e-4


@pytest.mark.parametrize("M,N,K,epi", [(200, 256, 192, "store"), (777, 384, 512, "store"), (12900, 4608, 1536, "store"),
                                       (12900, 8192, 1536, "swiglu"), (8229, 1536, 768, "gelu"), (3000, 4352, 1536, "bias_f32")])
def test_gemm_f16_build(M, N, K, epi):
    """The f16 build of the GEMM kernels (esmdiff_gemm_f16: the same sources compiled with IEEE-half operands, csrc/ed_half.h;
    128x128 and four-wave 256x256 kernels, every 16-bit epilogue): exact f16 products accumulated in f32, outputs rounded to f16 —
    |err| <= 2^-11 |ref| + accumulation noise, 1/8 of the bf16 bars; asymmetric-identity layout check; saturation instead of inf."""
    from esmdiff_amd import _native as Nn
    from esmdiff_amd.engine import gemm_f16
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g, device="cuda").half()
    W = (torch.randn(N, K, generator=g, device="cuda") / K ** 0.5).half()
    ref = A.float() @ W.float().t()
    if epi == "store":
        out = gemm_f16(A, W, Nn.EPI_BF16, alpha=0.866)
        assert out.dtype == torch.float16
        err = (out.float() - ref * 0.866).abs()
        assert float((err - ref.abs() * 2 ** -11).max()) < 5e-4, float(err.max())
    elif epi == "swiglu":
        out = gemm_f16(A, W, Nn.EPI_SWIGLU_BF16)
        r = ref.view(M, N // 64, 2, 32)
        want = (torch.nn.functional.silu(r[:, :, 0]) * r[:, :, 1]).reshape(M, N // 2)
        err = (out.float() - want).abs()
        assert float((err - want.abs() * 2 ** -10).max()) < 1.5e-3, float(err.max())
    elif epi == "gelu":
        bias = torch.randn(N, generator=g, device="cuda")
        out = gemm_f16(A, W, Nn.EPI_BIAS_GELU_BF16, bias=bias)
        want = torch.nn.functional.gelu(ref + bias)
        assert float((out.float() - want).abs().max()) < 4e-3 and float((out.float() - want).abs().mean()) < 3e-4
    else:
        nv, ld = 4101, 4104
        bias = torch.randn(N, generator=g, device="cuda")
        o = torch.full((M, ld), -7.0, device="cuda")
        gemm_f16(A, W, Nn.EPI_BIAS_F32, out=o, bias=bias, n_valid=nv)
        assert float((o[:, :nv] - (ref + bias)[:, :nv]).abs().max()) < 2e-4
    if (M, N, K) == (200, 256, 192):
        eye = torch.eye(256, device="cuda").half()
        Wa = (torch.arange(256 * 256, dtype=torch.float32, device="cuda").reshape(256, 256) % 251 - 125).half()
        assert torch.equal(gemm_f16(eye, Wa, Nn.EPI_BF16).float(), Wa.float().t())
        big = gemm_f16(torch.full((128, 64), 60000.0, device="cuda").half(), torch.full((128, 64), 2.0, device="cuda").half(), Nn.EPI_BF16)
        assert bool(torch.isfinite(big.float()).all()) and float(big.float().max()) == 65504.0      # saturates, never inf


@pytest.mark.parametrize("K", [768, 1536])
def test_gemm256w4_epilogues_large_m(K):
    """Every epilogue of the four-wave 256x256 kernel (the large-M default: >= 128 tiles) against f32 torch, ragged last
    row tile (M % 256 = 37), 12 and 24 K-tiles, the head's ragged 4101-of-4352 columns — the shapes the forward only
    reaches at benchmark size."""
    from esmdiff_amd import _native as Nn
    from esmdiff_amd.engine import gemm_bf16
    g = torch.Generator(device="cuda").manual_seed(K)
    M = 32 * 256 + 37
    A = _bf(torch.randn(M, K, generator=g, device="cuda"))
    W = _bf(torch.randn(1536, K, generator=g, device="cuda") / K ** 0.5)
    ref = A.float() @ W.float().t()
    out = gemm_bf16(A, W, Nn.EPI_BF16, alpha=0.866)
    err = (out.float() - ref * 0.866).abs()
    assert float((err - ref.abs() * 2 ** -8).max()) < 4e-3
    x0 = torch.randn(M, 1536, generator=g, device="cuda")
    x = x0.clone()
    gemm_bf16(A, W, Nn.EPI_RESID_F32, out=x, alpha=0.5)
    assert float((x - (x0 + ref * 0.5)).abs().max()) < 1e-3
    bias = torch.randn(1536, generator=g, device="cuda")
    out = gemm_bf16(A, W, Nn.EPI_BIAS_GELU_BF16, bias=bias)
    r2 = torch.nn.functional.gelu(ref + bias)
    e2 = (out.float() - r2).abs()                      # bf16 store: 2^-8 relative, plus the bf16-operand noise of the sum
    assert float((e2 - r2.abs() * 2 ** -7).max()) < 1e-2 and float(e2.mean()) < 2e-3
    H = 1024
    Wg = _bf(torch.randn(H, K, generator=g, device="cuda") / K ** 0.5)
    Wu = _bf(torch.randn(H, K, generator=g, device="cuda") / K ** 0.5)
    Wi = torch.stack([Wg.view(H // 32, 32, K), Wu.view(H // 32, 32, K)], 1).reshape(2 * H, K).contiguous()
    out = gemm_bf16(A, Wi, Nn.EPI_SWIGLU_BF16)
    r3 = torch.nn.functional.silu(A.float() @ Wg.float().t()) * (A.float() @ Wu.float().t())
    e3 = (out.float() - r3).abs()
    assert out.shape == (M, H) and float((e3 - r3.abs() * 2 ** -7).max()) < 1e-2 and float(e3.mean()) < 2e-3
    Wh = torch.zeros(4352, K, dtype=torch.bfloat16, device="cuda")
    Wh[:4101] = _bf(torch.randn(4101, K, generator=g, device="cuda") / K ** 0.5)
    bh = torch.zeros(4352, device="cuda")
    bh[:4101] = torch.randn(4101, generator=g, device="cuda")
    o = torch.full((M, 4104), -7.0, device="cuda")
    gemm_bf16(A, Wh, Nn.EPI_BIAS_F32, out=o, bias=bh, n_valid=4101)
    r4 = A.float() @ Wh[:4101].float().t() + bh[:4101]
    assert float((o[:, :4101] - r4).abs().max()) < 1e-3
    assert bool((o[-5:, :4101] != -7.0).all())                 # the ragged last rows were written


@pytest.mark.parametrize("M,K", [(240, 1536), (240, 4096), (517, 1536), (1000, 4096), (3, 1536)])
def test_branch_linear_layernorm_small_batch(tiny, M, K):
    """The small-batch residual branch: K-slice planes of the out-projection / FFN-down shapes summed inside the LayerNorm
    kernel (x += alpha * A W^T; y = LN(x) w + b) against f32 torch; S is the documented function of (N, K)."""
    _, _, eng, _, _ = tiny
    g = torch.Generator(device="cuda").manual_seed(M + K)
    D = 1536
    A = _bf(torch.randn(M, K, generator=g, device="cuda"))
    W = _bf(torch.randn(D, K, generator=g, device="cuda") / K ** 0.5)
    x0 = torch.randn(M, D, generator=g, device="cuda") * 3
    w, b = torch.randn(D, generator=g, device="cuda"), torch.randn(D, generator=g, device="cuda")
    for bias in (b, None):
        x = x0.clone()
        y, S = eng.branch_linear_layernorm(A, W, x, 0.866, w, bias)
        assert S == {1536: 4, 4096: 8}[K]
        xr = x0 + 0.866 * (A.float() @ W.float().t())
        assert float((x - xr).abs().max()) < 2e-4                       # f32 accumulation, only the summation order differs
        ref = torch.nn.functional.layer_norm(xr, (D,), w, bias, 1e-5)
        err = (y.float() - ref).abs()
        assert float((err - ref.abs() * 2 ** -8).max()) < 2e-3


def test_layernorm():
    from esmdiff_amd.engine import layernorm_bf16
    g = torch.Generator().manual_seed(2)
    for D in (512, 1536):
        x = (torch.randn(70, D, generator=g) * 3 + 0.5).cuda()
        w, b = torch.randn(D, generator=g).cuda(), torch.randn(D, generator=g).cuda()
        ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
        # f32 statistics; the only rounding is the bf16 store: |err| <= 2^-8 |ref| (+ f32 noise)
        err = (layernorm_bf16(x, w, b).float() - ref).abs()
        assert float((err - ref.abs() * 2 ** -8).max()) < 1e-4
        ref = torch.nn.functional.layer_norm(x, (D,), w, None, 1e-5)
        err = (layernorm_bf16(x, w, None).float() - ref).abs()
        assert float((err - ref.abs() * 2 ** -8).max()) < 1e-4


@pytest.mark.parametrize("B,L", [(2, 60), (1, 258), (3, 130), (2, 129), (2, 257), (2, 259), (1, 64), (1, 65), (2, 33), (2, 128),
                                 (1, 256), (1, 192), (2, 32), (1, 31), (1, 1)])
def test_attention_block(tiny, B, L):
    """q/k LayerNorm + rotary + softmax(QK^T/8)V against the oracle's MultiHeadAttentionRef internals."""
    cfg, sd, eng, net, _ = tiny
    attn = net.transformer.blocks[0].attn
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B, L, 3 * cfg.d_model, generator=g).to(torch.bfloat16)
    with torch.no_grad():
        q, k, v = torch.chunk(qkv.float(), 3, dim=-1)
        q, k = attn.q_ln(q), attn.k_ln(k)
        q, k = attn._rope(q, k)
        v = v.view(B, L, cfg.n_heads, 64).transpose(1, 2)
        ref = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v)
        ref = ref.transpose(1, 2).reshape(B * L, cfg.d_model)
    got = eng.attention(qkv.reshape(B * L, -1).contiguous().cuda(), attn.q_ln.weight.cuda(), attn.k_ln.weight.cuda(), B, L)
    err = (got.float().cpu() - ref).abs()
    # bf16 q,k,p,v operands with f32 accumulation: absolute error ~1e-2 on O(1) outputs
    assert float(err.max()) < 6e-2 and float(err.mean()) < 6e-3, (float(err.max()), float(err.mean()))


def test_attention_long_chain():
    """BASELINE config 4 shape: L_tok = 1026 (K/V per head exceed LDS -> tiled keys, online softmax)."""
    from esmdiff_amd.config import ModelConfig
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.weights import random_init_state_dict
    cfg = ModelConfig(d_model=512, n_heads=8, v_heads=32, n_layers=1)
    eng = Engine(cfg, random_init_state_dict(cfg, 3), max_batch=1, max_len=1026)
    B, L = 1, 1026
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, L, 3 * 512, generator=g) * 2).to(torch.bfloat16)
    ones = torch.ones(512)
    q, k, v = torch.chunk(qkv.float(), 3, -1)
    from oracle.esm3_ref import MultiHeadAttentionRef
    m = MultiHeadAttentionRef(512, 8)
    with torch.no_grad():
        m.q_ln.weight.fill_(1.0)
        m.k_ln.weight.fill_(1.0)
        qq, kk = m._rope(m.q_ln(q), m.k_ln(k))
        ref = torch.nn.functional.scaled_dot_product_attention(qq.transpose(1, 2), kk.transpose(1, 2),
                                                               v.view(B, L, 8, 64).transpose(1, 2))
        ref = ref.transpose(1, 2).reshape(B * L, 512)
    got = eng.attention(qkv.reshape(B * L, -1).contiguous().cuda(), ones.cuda(), ones.cuda(), B, L).float().cpu()
    err = (got - ref).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 6e-3
    eng.close()


# ---------------------------------------------------------------------------------------------------
# whole forward + sampling loop
@pytest.mark.parametrize("B,L", [(2, 60), (3, 258), (8, 110), (8, 290), (5, 258)])
def test_forward_logits_vs_oracle(tiny, B, L):
    # (8,110): 880 tokens, two streams of small-batch halves; (8,290): 2320 tokens, two streams on the regular path;
    # (5,258): 1290 tokens, one stream on the regular path — the engine's stream / path switches (engine.hip::forward)
    from esmdiff_amd.schedule import ddpm_schedule
    cfg, sd, eng, net, emb = tiny
    g = torch.Generator().manual_seed(L)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    x[:, 5:20] = torch.randint(0, 4096, (B, 15), generator=g)
    sch = ddpm_schedule(25)
    i = 6
    with torch.no_grad():
        cond = torch.tile(emb(sch.sigma_t[i] * torch.ones(B))[:, None, :], (1, L, 1))
        ref = net(structure_tokens=x, sequence_tokens=seq, auxiliary_embeddings=cond).structure_logits
    got = eng.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[i]).float().cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0)
    # bf16 GEMM operands through 2 blocks + head: logits (std ~0.6).  Bars = 2x the measured 0.015 / 0.0022 / 0.99
    assert float(cos) > 0.9999, float(cos)
    assert float(err.max()) < 0.03 and float(err.mean()) < 4.5e-3, (float(err.max()), float(err.mean()))
    assert float((got.argmax(-1) == ref.argmax(-1)).float().mean()) > 0.96      # measured 0.979 .. 1.0 over the five shapes


def test_forward_logits_f16_vs_oracle(tiny):
    """precision="f16" (the same kernel sources compiled with IEEE-half operands, csrc/ed_half.h) over the engine's stream / path
    switches: every operand rounding is 2^-12 instead of 2^-9, so the logit error is 1/8 of the bf16 engine's — measured
    0.0017 max / 0.00026 mean at production width; bars 2x that.  Sampling loop runs, a sample alone equals itself in a batch
    on the same dispatch path."""
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    cfg, sd, _, net, emb = tiny
    eng = Engine(cfg, sd, max_batch=8, max_len=300, precision="f16")
    sch = ddpm_schedule(25)
    i = 6
    for B, L in [(2, 60), (3, 258), (8, 110), (8, 290), (5, 258)]:
        g = torch.Generator().manual_seed(L)
        seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
        x = torch.full((B, L), MASK, dtype=torch.int64)
        x[:, 5:20] = torch.randint(0, 4096, (B, 15), generator=g)
        with torch.no_grad():
            cond = torch.tile(emb(sch.sigma_t[i] * torch.ones(B))[:, None, :], (1, L, 1))
            ref = net(structure_tokens=x, sequence_tokens=seq, auxiliary_embeddings=cond).structure_logits
        got = eng.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[i]).float().cpu()
        err = (got - ref).abs()
        assert float(err.max()) < 4e-3 and float(err.mean()) < 6e-4, (B, L, float(err.max()), float(err.mean()))
        assert float((got.argmax(-1) == ref.argmax(-1)).float().mean()) > 0.995
    out = eng.ddpm_sample(seq.cuda(), ddpm_schedule(3), seed=1).cpu()
    assert int((out == MASK).sum()) == 0
    eng.close()


def test_forward_with_coordinates_vs_oracle(monkeypatch):
    """Block 0's geometric attention (esmdiff_set_frames + geom.hip) against oracle/geom_ref.py inside the whole
    network: partly unknown coordinates (Inf = the inpainting marker, NaN at BOS/EOS), odd L, and the two-stream split.
    All-unknown coordinates must reproduce the unconditioned forward bit for bit (net.py:433-441)."""
    from esmdiff_amd.config import TINY
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.geometry import build_affine3d_from_coordinates
    from esmdiff_amd.weights import random_init_state_dict
    from oracle.esm3_ref import build_from_state_dict
    sd = random_init_state_dict(TINY, seed=3, with_geom=True)
    net, _ = build_from_state_dict(TINY, sd)
    B, L = 3, 71
    g = torch.Generator().manual_seed(11)
    ca = torch.cumsum(torch.randn(B, L, 3, generator=g) * 2.2, 1)
    xyz = torch.stack([ca + torch.randn(B, L, 3, generator=g) * 0.8, ca, ca + torch.randn(B, L, 3, generator=g) * 0.8], 2)
    xyz[:, 0] = float("nan")
    xyz[:, -1] = float("nan")
    xyz[:, 20:33] = float("inf")
    xyz[1, 40:] = float("inf")
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    x[:, 5:15] = torch.randint(0, 4096, (B, 10), generator=g)
    with torch.no_grad():
        ref = net(structure_tokens=x, sequence_tokens=seq, structure_coords=xyz).structure_logits
        ref0 = net(structure_tokens=x, sequence_tokens=seq).structure_logits
    assert float((ref - ref0).abs().max()) > 5e-2
    for streams in (1, 2):
        eng = Engine(TINY, sd, max_batch=B, max_len=L)
        eng.set_streams(streams, min_tokens=1)
        base = eng.forward_logits(x.cuda(), seq.cuda(), None).float().cpu().clone()
        eng.set_frames(*build_affine3d_from_coordinates(xyz))
        got = eng.forward_logits(x.cuda(), seq.cuda(), None).float().cpu().clone()
        eng.set_frames(*build_affine3d_from_coordinates(torch.full((B, L, 3, 3), float("nan"))))
        nan = eng.forward_logits(x.cuda(), seq.cuda(), None).float().cpu().clone()
        eng.set_frames(None)
        off = eng.forward_logits(x.cuda(), seq.cuda(), None).float().cpu().clone()
        with pytest.raises(RuntimeError):                      # frames for another shape
            eng.set_frames(*build_affine3d_from_coordinates(xyz[:2]))
            eng.forward_logits(x.cuda(), seq.cuda(), None)
        eng.close()
        err = (got - ref).abs()
        # same bar as the unconditioned forward: bf16 GEMM operands through 2 blocks + head
        assert float(err.max()) < 0.12 and float(err.mean()) < 1.2e-2, (streams, float(err.max()), float(err.mean()))
        assert float((got - base).abs().max()) > 5e-2          # the branch is live
        assert torch.equal(nan, base) and torch.equal(off, base)
    # weights without geom_attn: coordinates are refused, loudly
    eng = Engine(TINY, random_init_state_dict(TINY, seed=3), max_batch=B, max_len=L)
    with pytest.raises(RuntimeError, match="geom_attn"):
        eng.set_frames(*build_affine3d_from_coordinates(xyz))
    eng.close()


@pytest.mark.parametrize("B,L", [(3, 60), (2, 131)])
def test_structure_decoder_vs_oracle(B, L):
    """Structure tokens -> backbone coordinates (esmdiff_decoder_create / _decode: the ESM3 block stack at d = 768, the
    half-slab q/k LayerNorm path, then Dim6RotStructureHead) against oracle/decoder_ref.py."""
    from esmdiff_amd.config import TINY_DECODER
    from esmdiff_amd.engine import StructureDecoder
    from esmdiff_amd.weights import random_init_decoder_state_dict
    from oracle.decoder_ref import build_decoder_from_state_dict
    sd = random_init_decoder_state_dict(TINY_DECODER, seed=2)
    ref_net = build_decoder_from_state_dict(TINY_DECODER, sd)
    g = torch.Generator().manual_seed(L)
    tok = torch.randint(0, 4096, (B, L), generator=g)
    tok[:, 0], tok[:, -1] = 4098, 4097
    with torch.no_grad():
        ref = ref_net(tok)
    dec = StructureDecoder(TINY_DECODER, sd, max_batch=B, max_len=L)
    got, pl = dec.decode(tok.cuda(), return_plddt=True)
    got, pl = got.cpu(), pl.cpu()
    with torch.no_grad():
        _, pl_ref = ref_net(tok, return_plddt=True)
    # pLDDT = mean of the 50-bin categorical mixture (what to_pdb writes as B-factor): a softmax average, so the bf16
    # logit noise (~1e-2) moves it by well under a percent
    assert pl.shape == pl_ref.shape == (B, L - 2) and float(pl.min()) > 0 and float(pl.max()) < 1
    assert float((pl - pl_ref).abs().max()) < 5e-3, float((pl - pl_ref).abs().max())
    assert float(pl_ref.std()) > 1e-3                                       # the fixture is not constant
    assert got.shape == ref.shape == (B, L - 2, 3, 3)
    # the frame is rigid: ideal N-CA / CA-C bond lengths whatever the network says
    assert float(((got[:, :, 1] - got[:, :, 0]).norm(dim=-1) - 1.4592).abs().max()) < 2e-3
    assert float(((got[:, :, 1] - got[:, :, 2]).norm(dim=-1) - 1.5251).abs().max()) < 2e-3
    # translations are 10 x a head output computed from bf16 GEMM operands: a few hundredths of an Angstrom
    err = (got - ref).norm(dim=-1)
    assert float(err.mean()) < 0.08 and float(err.max()) < 0.6, (float(err.mean()), float(err.max()))
    # pairwise confidence head: predicted aligned error and pTM (decoder_output["ptm"], models/utils.py:73-76)
    tok2 = tok.clone()
    tok2[0, 5] = 4096                                                      # a special token inside: its pairs are masked out
    with torch.no_grad():
        ptm_ref, pae_ref = ref_net.confidence(tok2)
    _, ptm, pae = dec.decode(tok2.cuda(), return_ptm=True, return_pae=True)
    ptm, pae = ptm.cpu(), pae.cpu()
    assert ptm.shape == (B,) and pae.shape == (B, L, L)
    assert float(pae_ref[:, 1:-1, 1:-1].std()) > 1.0                       # the fixture's PAE is not flat
    assert float((pae[:, 0] - 16.0).abs().max()) < 1e-4 and float((pae[0, :, 5] - 16.0).abs().max()) < 1e-4   # masked: uniform
    e_pae = (pae - pae_ref).abs()
    assert float(e_pae.max()) < 0.6 and float(e_pae.mean()) < 0.05, (float(e_pae.max()), float(e_pae.mean()))
    assert float((ptm - ptm_ref).abs().max()) < 3e-3, (ptm, ptm_ref)
    assert 0 < float(ptm.min()) and float(ptm.max()) < 1
    with pytest.raises(RuntimeError, match="decoder"):
        dec._lib.esmdiff_forward_logits  # noqa: B018  (attribute exists)
        from esmdiff_amd import _native as Nn
        Nn.check(dec._lib.esmdiff_forward_logits(dec._h, None, None, None, None, 0, 1, 1, None), dec._h)
    dec.close()


@pytest.mark.parametrize("precision", ["bf16", "f32"])
@pytest.mark.parametrize("B,L", [(2, 40), (1, 131), (3, 9)])
def test_structure_encoder_vs_oracle(B, L, precision):
    """Coordinates -> structure tokens (esmdiff_encoder_create / _encode: kNN neighbourhoods, relative-position embedding,
    two geometric-attention + FFN blocks with biases, codebook lookup) against oracle/encoder_ref.py, with unknown
    residues; and what holds whatever esm's exact details are: tokens do not change under a rigid motion of the input."""
    from esmdiff_amd.config import TINY_ENCODER
    from esmdiff_amd.engine import StructureEncoder
    from esmdiff_amd.weights import random_init_encoder_state_dict
    from oracle.encoder_ref import build_encoder_from_state_dict
    sd = random_init_encoder_state_dict(TINY_ENCODER, seed=1)
    ref_net = build_encoder_from_state_dict(TINY_ENCODER, sd)
    g = torch.Generator().manual_seed(B * 100 + L)
    ca = torch.cumsum(torch.randn(B, L, 3, generator=g) * 2.2, 1)
    xyz = torch.stack([ca + torch.randn(B, L, 3, generator=g) * 0.8, ca, ca + torch.randn(B, L, 3, generator=g) * 0.8], 2)
    if L > 12:
        xyz[0, 5:8] = float("inf")
        xyz[-1, L - 2] = float("nan")
    with torch.no_grad():
        ref, _, d2 = ref_net(xyz, return_z=True)
    enc = StructureEncoder(TINY_ENCODER, sd, precision=precision)
    got = enc.encode(xyz).cpu()
    assert got.shape == ref.shape == (B, L)
    assert torch.equal(got == MASK, ref == MASK)
    if precision == "f32":     # the strict path: the very codes of the float32 oracle (a difference would need an exact tie)
        assert torch.equal(got, ref)
    # nearest-code lookups after bf16 GEMMs: the same code except where two codes are almost equally near IN THE
    # ORACLE'S OWN f32 distances — every disagreement must be such a near-tie, clear winners must agree
    live = ref != MASK
    d_ref = d2.gather(-1, ref.clamp(max=4095)[..., None])[..., 0]
    d_got = d2.gather(-1, got.clamp(max=4095)[..., None])[..., 0]
    rel = ((d_got - d_ref) / d_ref.clamp(min=1e-6))[live]
    two = d2.topk(2, dim=-1, largest=False)[0]
    gap12 = ((two[..., 1] - two[..., 0]) / two[..., 0].clamp(min=1e-6))[live]
    agree_m = (got == ref)[live]
    NOISE = 0.01
    assert float(rel.max()) < NOISE, float(rel.max())
    assert bool(agree_m[gap12 > 2 * NOISE].all())
    agree = float(agree_m.float().mean())
    assert agree > 0.85, agree
    assert len(set(ref.flatten().tolist())) > min(8, L // 2)            # the fixture is not degenerate
    A = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    if torch.det(A) < 0:
        A[:, 0] = -A[:, 0]
    moved = enc.encode(xyz @ A.T + torch.tensor([12.0, -40.0, 7.0])).cpu()
    assert float((moved == got).float().mean()) > 0.9
    enc.close()


def test_two_stream_forward_is_bitwise_identical(tiny, monkeypatch):
    """The engine runs the two halves of a large batch on two HIP streams (engine.hip::forward).  Samples are
    independent and every kernel's per-row arithmetic does not depend on the tiling, so logits and sampled ids must
    be bit-identical to the single-stream run, including odd B and repeated fork/join inside the sampling loop."""
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    cfg, sd, _, _, _ = tiny
    B, L = 5, 130
    g = torch.Generator().manual_seed(3)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
    x = torch.full((B, L), MASK, dtype=torch.int64)
    x[:, 3:40] = torch.randint(0, 4096, (B, 37), generator=g)
    x = x.cuda()
    sch = ddpm_schedule(6)
    outs = []
    for dual in (1, 2, 3):
        eng = Engine(cfg, sd, max_batch=B, max_len=L)
        eng.set_streams(dual, min_tokens=1)
        assert f"streams={dual}" in eng.describe_plan(B, L)
        lg = eng.forward_logits(x, seq, sch.t_freq[2]).clone()
        ids = eng.ddpm_sample(seq, sch, seed=5, sample_offset=0).clone()
        torch.cuda.synchronize()
        outs.append((lg.cpu(), ids.cpu()))
        eng.close()
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0])
        assert torch.equal(outs[0][1], o[1])


def test_ddpm_sample_end_to_end_vs_oracle(tiny):
    """Full loop on the device (esmdiff_ddpm_sample, Philox noise) vs the oracle driven step by step with
    the same noise: oracle forward (f32 torch) + C-oracle sampler.  bf16 logits differ from f32 logits in
    the low bits, so ids are compared by agreement rate after the FIRST update (identical inputs) and the
    whole run is checked for the structural invariants."""
    from esmdiff_amd.schedule import ddpm_schedule
    from oracle import c_oracle
    cfg, sd, eng, net, emb = tiny
    B, L, T = 3, 40, 5
    g = torch.Generator().manual_seed(8)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    sch = ddpm_schedule(T)
    out = eng.ddpm_sample(seq.cuda(), sch, seed=42, sample_offset=10).cpu()
    assert out.shape == (B, L) and int((out == MASK).sum()) == 0 and int(out.max()) <= 4100
    # first update, identical input state on both sides
    x0 = torch.full((B, L), MASK, dtype=torch.int64)
    with torch.no_grad():
        cond = torch.tile(emb(sch.sigma_t[0] * torch.ones(B))[:, None, :], (1, L, 1))
        ref_logits = net(structure_tokens=x0, sequence_tokens=seq, auxiliary_embeddings=cond).structure_logits
    want = c_oracle.ddpm_step(x0.numpy(), ref_logits.numpy(), sch.mc_t[0].item(), sch.mc_s[0].item(), seed=42,
                              sample_offset=10, step=0)
    lg = eng.forward_logits(x0.cuda(), seq.cuda(), sch.t_freq[0])
    got = eng.ddpm_step(x0.clone().cuda(), lg, sch.mc_t[0].item(), sch.mc_s[0].item(), seed=42, sample_offset=10,
                        step=0).cpu().numpy()
    agree = float((got == want).mean())
    assert agree > 0.98, agree          # 120 draws: at most 2 near-tie flips (measured 1.0)
    # and the sampler alone on the ENGINE's logits is bit-exact against the C oracle
    want2 = c_oracle.ddpm_step(x0.numpy(), lg.float().cpu().numpy(), sch.mc_t[0].item(), sch.mc_s[0].item(), seed=42,
                               sample_offset=10, step=0)
    assert np.array_equal(got, want2)
    # determinism + sharding independence of the whole loop
    again = eng.ddpm_sample(seq.cuda(), sch, seed=42, sample_offset=10).cpu()
    assert torch.equal(out, again)
    one = eng.ddpm_sample(seq[1:2].cuda(), sch, seed=42, sample_offset=11).cpu()
    assert torch.equal(one[0], out[1])
    # input_prior path (model.py:557-562): known tokens are carried through unchanged
    prior = torch.randint(0, 4096, (B, L), generator=g)
    prior[:, 0], prior[:, -1] = 4098, 4097
    prior[:, 7:15] = MASK
    o2 = eng.ddpm_sample(seq.cuda(), sch, seed=1, input_prior=prior.cuda()).cpu()
    keep = prior != MASK
    assert torch.equal(o2[keep], prior[keep]) and int((o2 == MASK).sum()) == 0


# ---------------------------------------------------------------------------------------------------
# drop-in surface: model wrapper, checkpoint loader, CLI
def test_model_wrapper_reference_rng_parity(tmp_path):
    """ddpm_sample(noise="torch-cpu") consumes torch's CPU uniform stream exactly like the reference
    (model.py:27): driving the ORACLE's torch sampler (restated model.py) around the engine's logits with the same
    seed gives the same ids at every position; and the checkpoint loader accepts the reference's 'module' layout."""
    from esmdiff_amd.config import TINY
    from esmdiff_amd.model import load_state_dict_from_lightning_ckpt
    from esmdiff_amd.schedule import timestep_embedding
    from esmdiff_amd.weights import random_init_state_dict
    from oracle import sampler_ref as R
    from types import SimpleNamespace
    sd = random_init_state_dict(TINY, seed=4)
    ck = tmp_path / "run" / "checkpoints" / "tiny.pt"
    ck.parent.mkdir(parents=True)
    torch.save({"module": sd}, ck)
    model = load_state_dict_from_lightning_ckpt(ck, device="cuda:0", max_batch=4, max_len=40, cfg=TINY)
    assert model.noise_removal is True
    with pytest.raises(FileNotFoundError):
        load_state_dict_from_lightning_ckpt(tmp_path / "nope.pt")
    B, L, T = 2, 20, 4
    g = torch.Generator().manual_seed(1)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    got = model.ddpm_sample(seq, num_steps=T, seed=123, noise="torch-cpu").cpu()

    class EngineNet:                     # the oracle sampler sees the ENGINE's logits (identical inputs)
        def __call__(self, structure_tokens=None, sequence_tokens=None, auxiliary_embeddings=None, labels=None):
            lg = model.net.forward_logits(structure_tokens.cuda(), sequence_tokens.cuda(), self.tf)
            return SimpleNamespace(structure_logits=lg.float().cpu().contiguous())

    net = EngineNet()

    class Emb:                            # records the sinusoid the engine needs; its MLP runs on the device
        def __call__(self, sigma):
            net.tf = timestep_embedding(sigma[:1].float(), TINY.freq_dim)[0]
            return torch.zeros(sigma.shape[0], TINY.d_model)

    ora = R.MDLMSamplerRef(net, Emb(), R.LogLinearNoiseRef(), True, True)
    torch.manual_seed(123)
    want = ora.ddpm_sample(seq, T)
    assert torch.equal(got, want)
    # error behaviour mirrored from model.py:556,562
    with pytest.raises(AssertionError):
        model.ddpm_sample(seq, num_steps=2, sample_max_t=0.5)
    with pytest.raises(AssertionError):
        model.ddpm_sample(seq, num_steps=2, input_prior=torch.zeros(B, L + 1, dtype=torch.int64))
    model.net.close()


def test_parity_stream_runs_across_batches_and_sigma0_conditioning():
    """(1) noise="torch-cpu" keeps ONE generator stream per run, like the reference's process-wide RNG: the second batch
    AND the next target (sample_offset 0 again) continue the stream — the reference never re-seeds — and only
    reset_parity_stream / a new seed restarts it.  (2) A model built with time_conditioning = false still adds sigma_embedder(0) (model.py:466-471, 535-541)."""
    import dataclasses
    from esmdiff_amd.config import TINY
    from esmdiff_amd.model import MaskedDiffusionLanguageModeling
    from esmdiff_amd.schedule import LogLinearNoise, timestep_embedding
    from esmdiff_amd.weights import random_init_state_dict
    from oracle.esm3_ref import build_from_state_dict
    sd = random_init_state_dict(TINY, seed=4)
    model = MaskedDiffusionLanguageModeling(sd, TINY, LogLinearNoise(), max_batch=4, max_len=40, device=0)
    B, L, T = 2, 20, 3
    g = torch.Generator().manual_seed(1)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    a0 = model.ddpm_sample(seq, num_steps=T, seed=9, sample_offset=0, noise="torch-cpu").cpu()
    a1 = model.ddpm_sample(seq, num_steps=T, seed=9, sample_offset=B, noise="torch-cpu").cpu()
    c0 = model.ddpm_sample(seq, num_steps=T, seed=9, sample_offset=0, noise="torch-cpu").cpu()   # "second target": runs on
    assert not torch.equal(a0, a1) and not torch.equal(c0, a0) and not torch.equal(c0, a1)
    model.reset_parity_stream(9)
    b0 = model.ddpm_sample(seq, num_steps=T, seed=9, sample_offset=0, noise="torch-cpu").cpu()
    b1 = model.ddpm_sample(seq, num_steps=T, seed=9, sample_offset=B, noise="torch-cpu").cpu()
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    d0 = model.ddpm_sample(seq, num_steps=T, seed=10, sample_offset=0, noise="torch-cpu").cpu()    # a new seed restarts too
    model.reset_parity_stream(10)
    assert torch.equal(d0, model.ddpm_sample(seq, num_steps=T, seed=10, noise="torch-cpu").cpu())
    model.net.close()
    # (2)
    cfg0 = dataclasses.replace(TINY, time_conditioning=False)
    m0 = MaskedDiffusionLanguageModeling(sd, cfg0, LogLinearNoise(), max_batch=4, max_len=40, device=0)
    net, emb = build_from_state_dict(cfg0, sd)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    with torch.no_grad():
        cond0 = torch.tile(emb(torch.zeros(B))[:, None, :], (1, L, 1))
        ref0 = net(structure_tokens=x, sequence_tokens=seq, auxiliary_embeddings=cond0).structure_logits
        refn = net(structure_tokens=x, sequence_tokens=seq).structure_logits
    tf = m0.net.conditioning_rows(timestep_embedding(torch.tensor([0.7, 0.3]), cfg0.freq_dim))
    assert tf is not None and torch.equal(tf[0], tf[1])                     # sigma is zeroed, the embedder still runs
    got = m0.net.forward_logits(x.cuda(), seq.cuda(), tf[0]).float().cpu()
    assert float((got - ref0).abs().max()) < 0.12 and float((ref0 - refn).abs().max()) > 0.12
    ids = m0.ddpm_sample(seq, num_steps=T, seed=1).cpu()                    # whole loop runs through the sigma(0) path
    assert int((ids == MASK).sum()) == 0
    m0.net.close()
    # out-of-range ids are refused by the host wrapper instead of being looked up
    eng = MaskedDiffusionLanguageModeling(sd, TINY, LogLinearNoise(), max_batch=2, max_len=24, device=0).net
    bad = seq.clone()
    bad[0, 3] = 64
    with pytest.raises(ValueError, match="out of range"):
        eng.forward_logits(x.cuda(), bad.cuda(), None)
    with pytest.raises(ValueError, match="out of range"):
        eng.forward_logits(torch.full((B, L), 4101, dtype=torch.int64).cuda(), seq.cuda(), None)
    eng.close()


def test_cli_ddpm_full_size_random_init(tmp_path):
    """The CLI end to end on ESM3-open-sized random weights: 58-residue synthetic target, 4 samples, 3 steps — at the default
    precision (certified: an f16 and an f32_split engine of the 1.4 B model; the run's json carries the certificate's counters)."""
    from esmdiff_amd.sample_esmdiff import main
    main(["--mode", "ddpm", "--random_init", "--synthetic_len", "58", "--num_samples", "4", "--num_steps", "3",
          "--output", str(tmp_path), "--no_timestamp", "--seed", "3"])
    out = tmp_path / "step3_eps1e-05_N4" / "synthetic58.tokens.npy"
    ids = np.load(out)
    assert ids.shape == (4, 58) and ids.min() >= 0 and ids.max() <= 4100 and (ids != 4096).all()
    import json
    meta = json.loads((tmp_path / "step3_eps1e-05_N4" / "synthetic58.json").read_text())
    assert meta["precision"] == "certified" and meta["certified"]["certificate"] == "k-sigma statistical + audit"
    assert meta["certified"]["audit_mismatches"] == 0


def test_sampler_whole_configs1_batch_vs_torch_sampler_with_rand_like(tiny):
    """the HIP sampler against the TORCH sampler — the reference's operation order (model.py:527-533, 602-607,
    24-28 as oracle/sampler_ref.py restates it and the goldens made by the reference's own code pin it), NOT the C oracle — with
    the uniforms torch.rand_like draws, on the whole configs[1] batch: 100 x 258 = 25 800 rows of 4 101 (105.8 M uniforms per
    update), three schedule points, ~20 % known rows.  Bar: every id equal."""
    from oracle.sampler_ref import MASK as M_, VOCAB, logits_parameterization_ref, sample_categorical_ref
    _, _, eng, _, _ = tiny
    B, L = 100, 258
    g = torch.Generator().manual_seed(31)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    known = torch.rand(B, L, generator=g) < 0.2
    x[known] = torch.randint(0, 4096, (int(known.sum()),), generator=g)
    total = flips = 0
    for i, (scale, mc_t, mc_s) in enumerate(((0.6, 0.999, 0.95904), (3.0, 0.5, 0.46004), (1.0, 0.04096, 1e-5))):
        logits = torch.randn(B, L, VOCAB, generator=g) * scale
        log_p = logits_parameterization_ref(logits, x)
        q = log_p.exp() * (torch.tensor(mc_t) - torch.tensor(mc_s))
        q[:, :, M_] = torch.tensor(mc_s)
        torch.manual_seed(1000 + i)
        drawn = sample_categorical_ref(q)                     # u = torch.rand_like(q), exactly the reference's call
        torch.manual_seed(1000 + i)
        u = torch.rand_like(q)                                # the same uniforms, to hand to the kernel
        keep = (x != MASK).to(x.dtype)
        want = keep * x + (1 - keep) * drawn
        pad = torch.zeros(B, L, 4104)
        pad[..., :VOCAB] = logits
        got = eng.ddpm_step(x.clone().cuda(), pad.cuda()[..., :VOCAB], mc_t, mc_s, u=u.cuda()).cpu()
        m = x == MASK
        total += int(m.sum())
        flips += int((got != want)[m].sum())
        assert torch.equal(got[~m], x[~m])
        del logits, log_p, q, u, pad
    print(f"HIP sampler vs torch sampler (rand_like uniforms): {flips} differing ids in {total} masked draws")
    assert total > 60_000 and flips == 0, (total, flips)


# ---------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5 at their full token counts (small model): size-independent properties
def test_config4_long_chain_batch_properties():
    """config 4: L_tok = 1026, num_samples = 32, 25 steps — the reference would split this into nine batches
    (sample_esmdiff.py:181-193); here it is one batch of 32 832 token rows through the tiled-key attention."""
    from esmdiff_amd.config import ModelConfig
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    from esmdiff_amd.weights import random_init_state_dict
    cfg = ModelConfig(d_model=512, n_heads=8, v_heads=32, n_layers=2)
    eng = Engine(cfg, random_init_state_dict(cfg, 5, device="cuda"), max_batch=32, max_len=1026)
    B, L, T = 32, 1026, 25
    g = torch.Generator().manual_seed(4)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
    sch = ddpm_schedule(T)
    out = eng.ddpm_sample(seq, sch, seed=9)
    assert out.shape == (B, L) and int((out == MASK).sum()) == 0 and int(out.max()) <= 4100
    assert torch.equal(out, eng.ddpm_sample(seq, sch, seed=9))                      # deterministic
    sub = eng.ddpm_sample(seq[:3], sch, seed=9, sample_offset=7)                    # rows 7..9 alone
    assert torch.equal(sub, out[7:10])                                             # batch / shard independent
    assert len({tuple(r.tolist()) for r in out[:, :64].cpu()}) == B                 # samples differ
    eng.close()


def test_config5_inpainting_prior_properties(tiny):
    """config 5 shape: L_tok = 258, 100 samples, 50 steps, a 64-position mask window (partial-mask resample):
    known tokens are carried through bit for bit, only the window is drawn."""
    from esmdiff_amd.config import TINY
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    from esmdiff_amd.weights import random_init_state_dict
    eng = Engine(TINY, random_init_state_dict(TINY, 6, device="cuda"), max_batch=100, max_len=258)
    B, L, T = 100, 258, 50
    g = torch.Generator().manual_seed(5)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
    prior = torch.randint(0, 4096, (1, L), generator=g).repeat(B, 1)
    prior[:, 0], prior[:, -1] = 4098, 4097
    prior[:, 97:161] = MASK                                   # residues 96..159 (token index = residue + 1)
    sch = ddpm_schedule(T)
    out = eng.ddpm_sample(seq, sch, seed=2, input_prior=prior.cuda()).cpu()
    keep = prior != MASK
    assert torch.equal(out[keep], prior[keep]) and int((out == MASK).sum()) == 0
    assert int((out[:, 97:161] < 4096).sum()) > 0.99 * B * 64
    half = eng.ddpm_sample(seq[:50], sch, seed=2, sample_offset=50, input_prior=prior[:50].cuda()).cpu()
    assert torch.equal(half, out[50:])
    eng.close()


# ---------------------------------------------------------------------------------------------------
# "gibbs" mode (entropy-ordered iterative unmasking; esm iterative_sampling_raw, parity unpinned vs esm)
@pytest.fixture(scope="module")
def tiny_stock():
    """The stock-ESM3 head shape: 4096-way structure head, no sigma embedder (what gibbs mode samples from without --ckpt)."""
    from esmdiff_amd.config import ModelConfig
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.weights import random_init_state_dict
    cfg = ModelConfig(d_model=512, n_heads=8, v_heads=128, n_layers=1, n_structure_heads=4096, time_conditioning=False)
    sd = {k: v for k, v in random_init_state_dict(cfg, seed=2).items() if not k.startswith("sigma_embedder.")}
    eng = Engine(cfg, sd, max_batch=4, max_len=300)
    yield eng
    eng.close()


@pytest.mark.parametrize("vocab", [4101, 4096])
@pytest.mark.parametrize("B,L,temp,top_p", [(2, 9, 1.4, 0.9), (3, 60, 0.7, 0.5), (2, 258, 1.0, 1.0)])
def test_gibbs_step_bit_exact(tiny, tiny_stock, vocab, B, L, temp, top_p):
    """Both head widths: entropy and nucleus over the whole row (4101 columns for the ESMDiff head), draws over the 4096
    codebook ids; some rows carry a heavy special-id logit so that the two widths really differ."""
    from oracle import c_oracle
    eng = tiny[2] if vocab == 4101 else tiny_stock
    assert eng.cfg.n_structure_heads == vocab
    g = torch.Generator().manual_seed(L)
    logits = torch.randn(B, L, 4104, generator=g) * 3
    logits[:, 1::4, 4099] += 9.0
    logits[0, 3, 4097] = 40.0                     # nothing valid survives the nucleus (4101-way): best valid id
    u = torch.rand(B, L, 4096, generator=g)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    x[:, 0], x[:, -1] = 4098, 4097
    x[0, 2] = 5
    n_un = torch.tensor([3, 1, 5][:B], dtype=torch.int32)
    want = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), temp, top_p, n_un.numpy(), u=u.numpy(), vocab=vocab)
    got = eng.gibbs_step(x.clone().cuda(), seq.cuda(), logits.cuda(), temp, top_p, n_un, u=u.cuda()).cpu().numpy()
    assert np.array_equal(got, want)
    want = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), temp, top_p, n_un.numpy(), seed=11,
                               sample_offset=2 ** 33 + 1, step=4, vocab=vocab)
    got = eng.gibbs_step(x.clone().cuda(), seq.cuda(), logits.cuda(), temp, top_p, n_un, seed=11,
                         sample_offset=2 ** 33 + 1, step=4).cpu().numpy()
    assert np.array_equal(got, want)
    assert ((got != x.numpy()).sum(1) == n_un.numpy()).all()
    assert int(got[(got != x.numpy())].max()) < 4096


def test_gibbs_step_fuzz_vs_c_oracle(tiny, tiny_stock):
    """30 random cases of the entropy-ordered unmasking step against the C oracle restatement, ids bit for bit: logit scales
    0.05-100 (flat rows: huge nuclei; peaked rows: a nucleus of one), quantised logits (ties in the nucleus cut and in the
    entropy ranking), temperatures 0 (arg-max) and 0.2-3, top-p 0.05-1.0, any number of positions to unmask including 0 and more than are
    masked, both head widths, both noise sources."""
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    for case in range(30):
        vocab = 4101 if case % 2 else 4096
        eng = tiny[2] if vocab == 4101 else tiny_stock
        B, L = int(rng.integers(1, 5)), int(rng.integers(3, 41))
        scale = float(rng.choice([0.05, 1.0, 10.0, 100.0]))
        z = rng.standard_normal((B, L, 4104)).astype(np.float32) * scale
        if case % 5 == 2:
            z = np.round(z / max(scale, 1e-3)).astype(np.float32) * np.float32(max(scale, 1e-3))
        if case % 3 == 0:
            z[:, ::3, 4099] += 9.0 * scale                                   # heavy special id: the two widths differ
        seq = np.concatenate([[0], rng.integers(4, 24, L - 2), [2]])[None].repeat(B, 0).astype(np.int64)
        x = np.full((B, L), MASK, dtype=np.int64)
        x[:, 0], x[:, -1] = 4098, 4097
        known = rng.random((B, L)) < float(rng.choice([0.0, 0.5, 0.95]))
        known[:, 0] = known[:, -1] = False
        x[known] = rng.integers(0, 4096, int(known.sum()))
        n_un = rng.integers(0, L + 2, B).astype(np.int32)
        temp, top_p = float(rng.choice([0.0, 0.2, 1.0, 1.4, 3.0])), float(rng.choice([0.05, 0.5, 0.9, 1.0]))   # 0: arg-max
        xt, st, zt, nt = torch.from_numpy(x).cuda(), torch.from_numpy(seq).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(n_un)
        if case % 4 == 0:
            u = rng.random((B, L, 4096), dtype=np.float32)
            want = c_oracle.gibbs_step(x, seq, z, temp, top_p, n_un, u=u, vocab=vocab)
            got = eng.gibbs_step(xt, st, zt, temp, top_p, nt, u=torch.from_numpy(u).cuda())
        else:
            seed, off, step = int(rng.integers(0, 2 ** 31)), int(rng.integers(0, 2 ** 40)), int(rng.integers(0, 500))
            want = c_oracle.gibbs_step(x, seq, z, temp, top_p, n_un, seed=seed, sample_offset=off, step=step, vocab=vocab)
            got = eng.gibbs_step(xt, st, zt, temp, top_p, nt, seed=seed, sample_offset=off, step=step)
        got = got.cpu().numpy()
        assert np.array_equal(got, want), (case, vocab, B, L, scale, temp, top_p, n_un.tolist())
        masked = (x == MASK).sum(1)
        assert ((got != x).sum(1) == np.minimum(n_un, masked)).all(), case
        assert np.array_equal(got[x != MASK], x[x != MASK])


def test_gibbs_iterative_sampling_raw_and_cli(tiny, tmp_path):
    """The reference's gibbs call shape: lists of ESMProtein / GenerationConfig in, list of ESMProtein out."""
    from esmdiff_amd.gibbs import iterative_sampling_raw, unmask_schedule
    from esmdiff_amd.sdk import ESMProtein, GenerationConfig
    from oracle import c_oracle
    cfg, sd, eng, net, emb = tiny
    seqs = "RPDFCLEPPYTGPCKARIIRYFYNAKAGLCQTFVYGGCRAKRNNFKSAEDCMRTCGGA"      # BPTI, data/targets/bpti
    prots = [ESMProtein(sequence=seqs) for _ in range(3)]
    cfgs = [GenerationConfig(track="structure", num_steps=16, temperature=1.4, top_p=0.9) for _ in range(3)]
    out = iterative_sampling_raw(eng, prots, cfgs, seed=5)
    assert len(out) == 3 and all(o.sequence == seqs and o.structure_tokens.shape == (58,) for o in out)
    toks = torch.stack([o.structure_tokens for o in out])
    assert int(toks.max()) < 4096 and int(toks.min()) >= 0
    again = iterative_sampling_raw(eng, prots[1:2], cfgs[1:2], seed=5, sample_offset=1)
    assert torch.equal(again[0].structure_tokens, toks[1])                  # shard independent
    # first step against the oracle chain: f32 oracle forward (no time conditioning) + C oracle step
    from esmdiff_amd.sdk import encode_sequence
    seq = encode_sequence(seqs)[None]
    x0 = torch.full((1, 60), MASK, dtype=torch.int64)
    x0[0, 0], x0[0, -1] = 4098, 4097
    k0 = unmask_schedule(58, 16)[0]
    lg = eng.forward_logits(x0.cuda(), seq.cuda(), None)
    got = eng.gibbs_step(x0.clone().cuda(), seq.cuda(), lg, 1.4, 0.9, torch.tensor([k0], dtype=torch.int32), seed=5).cpu()
    want = c_oracle.gibbs_step(x0.numpy(), seq.numpy(), lg.float().cpu().numpy(), 1.4, 0.9, np.array([k0], np.int32), seed=5)
    assert np.array_equal(got.numpy(), want) and int((got != x0).sum()) == k0
    # inpainting: known tokens stay, only masked residues are sampled; num_steps is clamped to #masked
    known = torch.randint(0, 4096, (58,))
    known[10:14] = MASK
    o2 = iterative_sampling_raw(eng, [ESMProtein(sequence=seqs, structure_tokens=known)],
                                [GenerationConfig(track="structure", num_steps=50, temperature=1.0, top_p=1.0)], seed=1)
    keep = known != MASK
    assert torch.equal(o2[0].structure_tokens[keep], known[keep]) and int((o2[0].structure_tokens == MASK).sum()) == 0
    with pytest.raises(NotImplementedError):
        iterative_sampling_raw(eng, prots[:1], [GenerationConfig(track="sequence")])
    # with a decoder attached the outputs are what the reference's call site needs: prot.to_pdb(tmp) works
    # (sample_esmdiff.py:124-128); without one the protein refuses, loudly
    from esmdiff_amd.config import TINY_DECODER
    from esmdiff_amd.engine import StructureDecoder
    from esmdiff_amd.weights import random_init_decoder_state_dict
    with pytest.raises(ValueError, match="decoder"):
        out[0].to_pdb(tmp_path / "no.pdb")
    dec = StructureDecoder(TINY_DECODER, random_init_decoder_state_dict(TINY_DECODER, seed=2), max_batch=2, max_len=60)
    o3 = iterative_sampling_raw(eng, prots, cfgs, seed=5, decoder=dec)
    assert all(torch.equal(a.structure_tokens, b.structure_tokens) for a, b in zip(o3, out))
    assert all(o.coordinates.shape == (58, 3, 3) and o.plddt.shape == (58,) for o in o3)
    assert all(o.ptm is not None and 0.0 < float(o.ptm) < 1.0 for o in o3)          # ESMProtein.ptm, from the pairwise head
    o3[1].to_pdb(tmp_path / "one.pdb")
    back = ESMProtein.from_pdb(tmp_path / "one.pdb")
    assert back.sequence == seqs and float((back.coordinates - o3[1].coordinates).abs().max()) < 1e-3
    bf = [float(ln[60:66]) for ln in (tmp_path / "one.pdb").read_text().splitlines() if ln.startswith("ATOM")]
    assert abs(bf[0] - float(o3[1].plddt[0])) < 6e-3
    dec.close()


def test_gibbs_strategy_random_and_invalid_ids(tiny):
    """GenerationConfig.strategy = "random" / .invalid_ids (r03) on the device: the step is bit-exact vs the C oracle with the
    same options (both noise sources for invalid_ids; Philox for the random position keys), explicit uniforms + random is
    refused, and iterative_sampling_raw honours both fields end to end and restores the defaults afterwards."""
    from esmdiff_amd.gibbs import iterative_sampling_raw
    from esmdiff_amd.sdk import ESMProtein, GenerationConfig
    from oracle import c_oracle
    eng = tiny[2]
    B, L = 3, 60
    g = torch.Generator().manual_seed(77)
    logits = torch.randn(B, L, 4104, generator=g) * 3
    u = torch.rand(B, L, 4096, generator=g)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    x[:, 0], x[:, -1] = 4098, 4097
    n_un = torch.tensor([7, 1, 12], dtype=torch.int32)
    inv = sorted({int(v) for v in logits[..., :4096].argmax(-1).flatten().tolist()}) + [4100]   # every row's arg-max is forbidden
    try:
        eng.set_gibbs_options("entropy", inv)
        want = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 0.3, 0.9, n_un.numpy(), u=u.numpy(), vocab=4101, invalid_ids=inv)
        got = eng.gibbs_step(x.clone().cuda(), seq.cuda(), logits.cuda(), 0.3, 0.9, n_un, u=u.cuda()).cpu().numpy()
        assert np.array_equal(got, want) and not np.isin(got[got != x.numpy()], inv).any()
        eng.set_gibbs_options("random", inv)
        want = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un.numpy(), seed=4, sample_offset=9, step=2,
                                   vocab=4101, invalid_ids=inv, strategy="random")
        got = eng.gibbs_step(x.clone().cuda(), seq.cuda(), logits.cuda(), 1.4, 0.9, n_un, seed=4, sample_offset=9, step=2).cpu().numpy()
        assert np.array_equal(got, want) and ((got != x.numpy()).sum(1) == n_un.numpy()).all()
        ent = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un.numpy(), seed=4, sample_offset=9, step=2,
                                  vocab=4101, invalid_ids=inv)
        assert not np.array_equal(got != x.numpy(), ent != x.numpy())           # other positions than the entropy order picks
        with pytest.raises(RuntimeError, match="Philox"):
            eng.gibbs_step(x.clone().cuda(), seq.cuda(), logits.cuda(), 1.4, 0.9, n_un, u=u.cuda())
        with pytest.raises(RuntimeError, match="not a structure-track id"):
            eng.set_gibbs_options("entropy", [5000])
        with pytest.raises(ValueError):
            eng.set_gibbs_options("greedy")
    finally:
        eng.set_gibbs_options()
    base = c_oracle.gibbs_step(x.numpy(), seq.numpy(), logits.numpy(), 1.4, 0.9, n_un.numpy(), seed=4, vocab=4101)
    assert np.array_equal(eng.gibbs_step(x.clone().cuda(), seq.cuda(), logits.cuda(), 1.4, 0.9, n_un, seed=4).cpu().numpy(), base)
    # end to end through the reference's call shape
    prots = [ESMProtein(sequence="RPDFCLEPPYTGPCKARIIRYFYNAKAGLCQTFVYGGCRAKRNNFKSAEDCMRTCGGA")] * 2
    mk = lambda **kw: [GenerationConfig(track="structure", num_steps=6, temperature=1.4, top_p=0.9, **kw)] * 2   # noqa: E731
    a = iterative_sampling_raw(eng, prots, mk(), seed=3)
    r = iterative_sampling_raw(eng, prots, mk(strategy="random"), seed=3)
    banned = sorted({int(v) for p in a for v in p.structure_tokens.tolist()})[:40]
    v = iterative_sampling_raw(eng, prots, mk(invalid_ids=banned), seed=3)
    assert all(int((p.structure_tokens == MASK).sum()) == 0 and int(p.structure_tokens.max()) < 4096 for p in a + r + v)
    assert not torch.equal(a[0].structure_tokens, r[0].structure_tokens)
    assert not any(np.isin(p.structure_tokens.numpy(), banned).any() for p in v)
    again = iterative_sampling_raw(eng, prots, mk(), seed=3)                      # the options did not leak into the next call
    assert all(torch.equal(p.structure_tokens, q.structure_tokens) for p, q in zip(a, again))


def test_cli_gibbs_default_mode(tmp_path):
    from esmdiff_amd.sample_esmdiff import main
    main(["--random_init", "--synthetic_len", "40", "--num_samples", "3", "--num_steps", "8", "--output", str(tmp_path),
          "--no_timestamp", "--seed", "2", "--precision", "bf16"])         # --mode defaults to gibbs, as in the reference (the default
                                                                           # precision, certified, at full size: test_cli_ddpm_full_size_random_init)
    ids = np.load(tmp_path / "T1.4_step8_topp0.9_N3" / "synthetic40.tokens.npy")
    assert ids.shape == (3, 40) and ids.min() >= 0 and ids.max() < 4096


def test_cli_gibbs_with_stock_esm3_state_dict(tmp_path):
    """Without --ckpt the reference samples from the stock ESM3 (4096-way head, no time conditioning, plain esm keys:
    sample_esmdiff.py:37, :252-255).  Here the state dict comes from a file (--esm3_ckpt); ddpm mode is refused as there."""
    from esmdiff_amd.config import ModelConfig
    from esmdiff_amd.sample_esmdiff import main
    from esmdiff_amd.weights import random_init_state_dict
    cfg = ModelConfig(n_layers=2, n_structure_heads=4096, time_conditioning=False)      # ESM3-open width, two blocks
    sd = {k[len("net."):]: v for k, v in random_init_state_dict(cfg, seed=0).items() if k.startswith("net.")}
    torch.save(sd, tmp_path / "esm3.pt")
    import esmdiff_amd.config as C
    old = C.ESM3_OPEN_STOCK
    C.ESM3_OPEN_STOCK = cfg
    try:
        main(["--esm3_ckpt", str(tmp_path / "esm3.pt"), "--synthetic_len", "24", "--num_samples", "2", "--num_steps", "4",
              "--output", str(tmp_path), "--no_timestamp"])
        with pytest.raises(AssertionError, match="Only Gibbs"):
            main(["--esm3_ckpt", str(tmp_path / "esm3.pt"), "--synthetic_len", "24", "--mode", "ddpm", "--output", str(tmp_path)])
    finally:
        C.ESM3_OPEN_STOCK = old
    ids = np.load(tmp_path / "T1.4_step4_topp0.9_N2" / "synthetic24.tokens.npy")
    assert ids.shape == (2, 24) and ids.min() >= 0 and ids.max() < 4096


def test_cli_writes_multi_model_pdb_with_decoder(tmp_path):
    """The reference's artefact: <basename>.pdb with one MODEL per sample (sample_esmdiff.py:225-231), produced by the
    decoder engine (random weights here) + merge_pdbfiles; readable back with the in-tree PDB reader."""
    from esmdiff_amd.sample_esmdiff import main
    main(["--random_init", "--tiny", "--random_init_decoder", "--synthetic_len", "30", "--mode", "ddpm", "--num_samples", "4",
          "--num_steps", "4", "--output", str(tmp_path), "--no_timestamp"])
    d = tmp_path / "step4_eps1e-05_N4"
    ids = np.load(d / "synthetic30.tokens.npy")
    text = (d / "synthetic30.pdb").read_text().splitlines()
    assert ids.shape == (4, 30)
    # (the reference's merge closes single-model inputs twice at the very end — eval_utils.py:437-492, golden g8)
    assert sum(l.startswith("MODEL") for l in text) == 4 and sum(l.startswith("ENDMDL") for l in text) in (4, 5)
    assert text[-1].startswith("END") and all(len(l) == 80 for l in text)
    assert sum(l.startswith("ATOM") for l in text) == 4 * (30 * 3 + 29)     # N, CA, C + the inferred O (none on the last residue)
    ca = [l for l in text if l.startswith("ATOM") and l[12:16].strip() == "CA"]
    assert len(ca) == 120
    import json
    meta = json.loads((d / "synthetic30.json").read_text())             # pTM of every sample, from the decoder's pairwise head
    assert len(meta["ptm"]) == 4 and all(0.0 < v < 1.0 for v in meta["ptm"])
    # the default (gibbs) driver writes them too
    main(["--random_init", "--tiny", "--random_init_decoder", "--synthetic_len", "30", "--num_samples", "3", "--num_steps", "4",
          "--output", str(tmp_path), "--no_timestamp"])
    meta = json.loads((tmp_path / "T1.4_step4_topp0.9_N3" / "synthetic30.json").read_text())
    assert len(meta["ptm"]) == 3 and all(0.0 < v < 1.0 for v in meta["ptm"])


def test_cli_gibbs_inpainting_from_pdb(tmp_path):
    """`--mask_ids` in the default mode, as the reference runs it (sample_esmdiff.py:282-300, :88-96): the input PDB's
    backbone conditions the model (geometric attention), the masked residues lose sequence and coordinates, every
    structure token is sampled.  The conditioning must matter: other coordinates, other samples, same seed."""
    from esmdiff_amd.pdbio import write_backbone_pdb
    from esmdiff_amd.sample_esmdiff import main
    seq = "RPDFCLEPPYTGPCKARIIRYFYNAKAGLCQTFVYGGCRA"
    g = np.random.default_rng(0)
    outs = []
    for k in range(2):
        ca = np.cumsum(g.normal(size=(len(seq), 3)) * 2.2, 0)
        xyz = np.stack([ca + g.normal(size=ca.shape) * 0.8, ca, ca + g.normal(size=ca.shape) * 0.8], 1)
        d = tmp_path / f"in{k}"
        d.mkdir()
        write_backbone_pdb(d / "toy.pdb", seq, xyz)
        main(["--random_init", "--input", str(d), "--num_samples", "3", "--num_steps", "6", "--mask_ids", "10,11,12,13",
              "--output", str(tmp_path / f"out{k}"), "--no_timestamp", "--seed", "4", "--tiny"])
        ids = np.load(tmp_path / f"out{k}" / "T1.4_step6_topp0.9_N3" / "toy.tokens.npy")
        assert ids.shape == (3, len(seq)) and ids.min() >= 0 and ids.max() < 4096
        outs.append(ids)
    assert (outs[0] != outs[1]).mean() > 0.3
    with pytest.raises(SystemExit, match="encoder"):
        main(["--random_init", "--input", str(tmp_path / "in0"), "--mode", "ddpm", "--mask_ids", "1,2", "--output",
              str(tmp_path / "o"), "--tiny"])
    # DDPM inpainting as the reference does it (sample_esmdiff.py:166-201): the encoder tokenises the known backbone, the
    # masked residues (and, by the reference's token-space quirk, their left neighbours) are re-sampled, the rest is kept
    from esmdiff_amd.config import TINY_ENCODER
    from esmdiff_amd.engine import StructureEncoder
    from esmdiff_amd.pdbio import read_pdb_backbone
    from esmdiff_amd.weights import random_init_encoder_state_dict
    main(["--random_init", "--tiny", "--random_init_encoder", "--input", str(tmp_path / "in0"), "--mode", "ddpm",
          "--mask_ids", "10,11,12,13", "--num_samples", "3", "--num_steps", "5", "--output", str(tmp_path / "od"),
          "--no_timestamp", "--seed", "4"])
    ids = np.load(tmp_path / "od" / "step5_eps1e-05_N3" / "toy.tokens.npy")
    _, xyz = read_pdb_backbone(tmp_path / "in0" / "toy.pdb")
    enc = StructureEncoder(TINY_ENCODER, random_init_encoder_state_dict(TINY_ENCODER, seed=4, device="cuda:0"))
    xyz = torch.from_numpy(xyz).float()
    xyz[10:14] = float("inf")              # protseq_to_data removes the masked residues' coordinates before encoding
    known = enc.encode(xyz[None])[0].cpu().numpy()
    enc.close()
    keep = np.ones(len(seq), bool)
    keep[9:14] = False                     # residues 10-13 (coordinates removed) and residue 9 (token index 10)
    assert ids.shape == (3, len(seq)) and (ids[:, keep] == known[keep]).all() and ids.max() < 4096


# ---------------------------------------------------------------------------------------------------
# error behaviour of the boundary (status codes -> RuntimeError; the reference: assert / strict load_state_dict)
def test_engine_error_behaviour(tiny):
    from esmdiff_amd.config import TINY
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    from esmdiff_amd.weights import random_init_state_dict
    cfg, sd, eng, _, _ = tiny
    # strict weight table (checkpoint_utils.py:64 load_state_dict): a missing or mis-shaped tensor is an error
    bad = dict(sd)
    del bad["net.transformer.blocks.1.ffn.3.weight"]
    with pytest.raises(RuntimeError, match="missing weight 'transformer.blocks.1.ffn.3.weight'"):
        Engine(TINY, bad, 2, 16)
    bad = dict(sd)
    bad["net.transformer.norm.weight"] = torch.ones(7)
    with pytest.raises(RuntimeError, match="transformer.norm.weight"):
        Engine(TINY, bad, 2, 16)
    # capacity: engine was created for max_batch=8, max_len=300
    seq = torch.zeros(9, 20, dtype=torch.int64).cuda()
    with pytest.raises(RuntimeError, match="capacity"):
        eng.ddpm_sample(seq, ddpm_schedule(2), seed=0)
    with pytest.raises(RuntimeError, match="capacity"):
        eng.forward_logits(torch.zeros(1, 301, dtype=torch.int64).cuda(), torch.zeros(1, 301, dtype=torch.int64).cuda(), None)
    # noise source is mandatory for a sampling step
    lg = torch.zeros(1, 4, 4104, device="cuda")
    with pytest.raises(ValueError):
        eng.ddpm_step(torch.full((1, 4), MASK, dtype=torch.int64).cuda(), lg, 0.5, 0.4)
    with pytest.raises(RuntimeError, match="temperature"):
        eng.gibbs_step(torch.full((1, 4), MASK, dtype=torch.int64).cuda(), torch.zeros(1, 4, dtype=torch.int64), lg, -0.5, 0.9,
                       torch.ones(1, dtype=torch.int32), seed=1)                 # (0 is arg-max decoding; negative is refused)
    # the engine keeps working after errors
    out = eng.ddpm_sample(torch.tensor([[0, 5, 6, 7, 2]]).cuda(), ddpm_schedule(2), seed=0)
    assert out.shape == (1, 5) and int((out == MASK).sum()) == 0


def test_bench_line_contract(tmp_path):
    """bench.py's single JSON line as the driver reads it (small model, seconds): every field of the contract, the roofline
    and cpu_baseline objects, and that nothing else is printed on stdout."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--tiny", "--steps", "2", "--warmup", "1",
                        "--samples-per-gpu", "6", "--residues", "40"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["dtype"] == "bf16" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 6 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-2          # value = samples / time of the region
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and isinstance(cb["sample"], str)
    assert "power" in d                                                     # hwmon sample or null, never missing


def test_f16_engine_saturates_instead_of_overflowing(tiny):
    """f16 has 5 exponent bits: every f32 -> f16 conversion of the ed16 kernels clamps to +-65504 (csrc/ed_half.h), so an outlier
    activation costs accuracy, never an inf / NaN that would poison the residual stream.  One block's FFN-up weight is scaled by 400
    (SwiGLU products far beyond 65504): the f16 engine's logits stay finite and the sampling loop still completes; the bf16 engine,
    which has the range, is the reference for 'finite'."""
    from esmdiff_amd.engine import Engine
    from esmdiff_amd.schedule import ddpm_schedule
    cfg, sd, _, _, _ = tiny
    sd2 = dict(sd)
    key = next(k for k in sd2 if k.endswith("blocks.1.ffn.1.weight"))
    sd2[key] = sd2[key] * 400.0
    B, L = 2, 40
    g = torch.Generator().manual_seed(2)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
    x = torch.full((B, L), MASK, dtype=torch.int64)
    sch = ddpm_schedule(3)
    for prec in ("bf16", "f16"):
        eng = Engine(cfg, sd2, max_batch=B, max_len=L, precision=prec)
        lg = eng.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[0]).float()
        assert bool(torch.isfinite(lg).all()), prec
        out = eng.ddpm_sample(seq.cuda(), sch, seed=1).cpu()
        assert int((out == MASK).sum()) == 0, prec
        eng.close()


@pytest.mark.gpu
def test_encode_decode_round_trip_call_shape(tmp_path):
    """`encode_decode(model, pdb)` (models/utils.py:166-194) on the encoder + decoder engines: PDB -> tokens -> backbone, the pair
    (coords, coords_pred) returned; same result from a path and from an ESMProtein; deterministic.  (Random-init VQ-VAE weights:
    the reconstruction itself means nothing here — the real checkpoints' does.)"""
    from esmdiff_amd.config import TINY_DECODER, TINY_ENCODER
    from esmdiff_amd.engine import StructureDecoder, StructureEncoder
    from esmdiff_amd.pdbio import write_backbone_pdb
    from esmdiff_amd.sdk import ESMProtein, encode_decode
    from esmdiff_amd.weights import random_init_decoder_state_dict, random_init_encoder_state_dict
    seq = "RPDFCLEPPYTGPCKARIIRYFYNAKAGLCQTFVYGGCRA"
    g = np.random.default_rng(1)
    ca = np.cumsum(g.normal(size=(len(seq), 3)) * 2.2, 0)
    xyz = np.stack([ca + g.normal(size=ca.shape) * 0.8, ca, ca + g.normal(size=ca.shape) * 0.8], 1).astype(np.float32)
    write_backbone_pdb(tmp_path / "toy.pdb", seq, xyz)
    enc = StructureEncoder(TINY_ENCODER, random_init_encoder_state_dict(TINY_ENCODER, seed=3))
    dec = StructureDecoder(TINY_DECODER, random_init_decoder_state_dict(TINY_DECODER, seed=2), max_batch=1, max_len=len(seq) + 2)
    coords, pred = encode_decode(enc, dec, tmp_path / "toy.pdb")
    assert coords.shape == pred.shape == (len(seq), 3, 3) and bool(torch.isfinite(pred).all())
    assert np.allclose(coords.numpy(), xyz, atol=6e-4)                                   # what the PDB's %8.3f records hold
    c2, p2 = encode_decode(enc, dec, ESMProtein(sequence=seq, coordinates=torch.from_numpy(np.round(xyz, 3))))
    assert torch.equal(p2, encode_decode(enc, dec, tmp_path / "toy.pdb")[1]) or torch.allclose(p2, pred, atol=1e-3)
    with pytest.raises(ValueError, match="Invalid input type"):
        encode_decode(enc, dec, 3)
    with pytest.raises(ValueError, match="no coordinates"):
        encode_decode(enc, dec, ESMProtein(sequence=seq))
    dec.close()
