"""TEST INFRASTRUCTURE — float64 restatement of the certificate of esmdiff_gibbs_step_rows (csrc/gibbs.hip, MARGIN kernels):
for each prompt of a batch, could logits that are only known up to an error have decided this step otherwise?

No reference counterpart (the reference has one precision; esm's iterative_sampling_raw, which the step itself restates, is
reached at /root/reference/slm/sample_esmdiff.py:114-122).  The ids come from the C oracle's plain step (oracle_gibbs_step, one
prompt at a time with that prompt's own Philox index, step and count — the contract of the per-prompt entry); the report is
recomputed here from the definitions, not from the kernel's search:

  R   bound on the error of the difference of any two logits of a row  ->  every probability is known up to exp(+-R)
  E   bound on the error of a row's entropy
  a token v is SURELY kept    when  mass{z_j >= z_v - R} * exp(R) <= top_p        (or it is the only token within R of the maximum)
              SURELY dropped  when  mass{z_j >= z_v + R} * exp(-R) > top_p  and  z_v + R < max z
  bit 0 (race)     the winner does not lead the best other possibly-kept valid token by more than exp(R / temperature)
                   (temperature 0: by more than R in the logit)
  bit 1 (nucleus)  the winner is not surely kept, or a possibly-kept token beats it, or the fall-back was taken
  bit 2 (order)    largest selected entropy and smallest unselected eligible entropy are not more than 2 E apart

Only the rows a prompt unmasks in the step count.  Used by tests/test_certified_cpu.py (stand-in engines) and by
tests/test_gpu_kernels.py (the kernel's flags must lie between this report at slightly smaller and slightly larger bounds).
"""
from __future__ import annotations

import numpy as np

from . import c_oracle

MASK, NVALID = 4096, 4096
GIBBS_STEP_DTYPE = [("sample_index", "<u8"), ("step", "<i4"), ("n_unmask", "<i4")]


def row_report(z, u, temperature, top_p, R, winner, invalid=()):
    """z (vocab,) float64 logits of one row, u (4096,) its uniforms, winner = the id the plain step drew.
    Returns (flag bits 0 / 1, race gap in logit units)."""
    m = z.max()
    p = np.exp(z - m)
    p /= p.sum()
    order = np.argsort(-z, kind="stable")
    zs, cs = z[order], np.cumsum(p[order])

    def mass_ge(theta):                       # probability mass of {j : z_j >= theta}
        n = int(np.searchsorted(-zs, -theta, side="right"))
        return cs[n - 1] if n > 0 else 0.0

    V = z.shape[0]
    if top_p >= 1.0:
        sure = np.ones(V, bool)
        maybe = np.ones(V, bool)
    else:
        near = int((z >= m - R).sum())
        sure = np.array([mass_ge(z[v] - R) * np.exp(R) <= top_p or (z[v] == m and near == 1) for v in range(V)])
        maybe = np.array([not (mass_ge(z[v] + R) * np.exp(-R) > top_p and z[v] + R < m) for v in range(V)])
    valid = np.zeros(V, bool)
    valid[:NVALID] = True
    for v in invalid:
        valid[int(v)] = False
    cand = np.nonzero(maybe & valid)[0]
    if temperature > 0:
        g = 1e-10 - np.log(u.astype(np.float64) + 1e-10)
        val = np.exp((z[:NVALID] - m) / temperature) / g
    else:
        val = z[:NVALID] - m
    if len(cand) == 0:
        return 3, 0.0
    cv = val[cand]
    o = np.argsort(-cv, kind="stable")
    c1_i, c1 = int(cand[o[0]]), cv[o[0]]
    c2 = cv[o[1]] if len(cand) > 1 else None
    if c2 is None:
        gap, race_ok = np.inf, True
    elif temperature > 0:
        gap = (np.log(c1) - np.log(max(c2, 1e-300))) * temperature
        race_ok = c1 > c2 * np.exp(R / temperature)
    else:
        gap = c1 - c2
        race_ok = gap > R
    nuc_ok = c1_i == winner and bool(sure[winner])
    return (0 if race_ok else 1) | (0 if nuc_ok else 2), float(max(gap, 0.0))


def gibbs_step_rows(x, seq, logits, temperature, top_p, params, seed, R=None, E=None, vocab=4101, strategy="entropy",
                    invalid_ids=()):
    """x, seq (B, L) int64; logits (B, L, >= vocab) float32; params: structured array of GIBBS_STEP_DTYPE (one per prompt).
    Returns (new x, flags (B,) int32, gaps (B, 2) float32) — flags / gaps are None without bounds."""
    x = np.array(x, dtype=np.int64, copy=True)
    seq = np.asarray(seq, dtype=np.int64)
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    B, L = x.shape
    margins = R is not None
    flags = np.zeros(B, np.int32) if margins else None
    gaps = np.full((B, 2), np.inf, np.float32) if margins else None
    for b in range(B):
        si, stp, k = int(params["sample_index"][b]), int(params["step"][b]), int(params["n_unmask"][b])
        if k <= 0:
            continue
        xb = x[b:b + 1]
        new, ent, smp = c_oracle.gibbs_step(xb, seq[b:b + 1], lg[b:b + 1], temperature, top_p, np.array([k], np.int32), seed=seed,
                                            sample_offset=si, step=stp, return_aux=True, vocab=vocab, strategy=strategy,
                                            invalid_ids=invalid_ids)
        sel = np.nonzero(new[0] != xb[0])[0]
        x[b] = new[0]
        if not margins:
            continue
        f, g = 0, np.inf
        for l in sel:
            u = c_oracle.philox_uniforms(seed, si, stp, int(l), NVALID)
            rf, rg = row_report(lg[b, l, :vocab].astype(np.float64), u, temperature, top_p, R, int(new[0, l]), invalid_ids)
            f |= rf
            g = min(g, rg)
        s = seq[b]
        eligible = (xb[0] == MASK) & (s != 0) & (s != 1) & (s != 2)
        uns = eligible.copy()
        uns[sel] = False
        hgap = np.inf
        if len(sel) and uns.any():
            hgap = float(ent[0][uns].min()) - float(ent[0][sel].max())
            if strategy == "entropy" and not (hgap > 2.0 * E):
                f |= 4
        flags[b], gaps[b, 0], gaps[b, 1] = f, min(g, 3.4e38), min(hgap, 3.4e38)
    return x, flags, gaps
