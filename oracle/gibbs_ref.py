"""TEST INFRASTRUCTURE — torch restatement of the "gibbs" sampler the reference reaches through
esm.utils.generation.iterative_sampling_raw (/root/reference/slm/sample_esmdiff.py:114-122).

PARITY UNPINNED: esm==3.0.4 (requirements.txt:30) is not vendored in the reference and not installable here; the
semantics below are SURVEY.md Appendix B, written from the published package's behaviour.  This file states them with
plain torch ops (sort-based nucleus, log_softmax entropy, top-k by entropy) and is used to cross-check the canonical
C oracle / HIP kernel, which implement the same rule without a sort.
"""
from __future__ import annotations

import math
from typing import List

import torch

MASK = 4096
NVALID = 4096


def unmask_schedule(total_to_sample: int, num_steps: int) -> List[int]:
    """Positions to unmask at each step: cosine schedule, num_steps clamped to the number of masked positions."""
    T = min(num_steps, total_to_sample)
    out, still = [], total_to_sample
    for t in range(T):
        after = int(math.cos((t + 1) / T * math.pi / 2) * total_to_sample + 0.1)
        k = max(still - after, 0)
        out.append(k)
        still -= k
    return out


def top_p_logits(logits: torch.Tensor, top_p: float) -> torch.Tensor:
    """Keep the sorted prefix whose cumulative probability is <= top_p, always the top-1; others -> -inf."""
    if top_p >= 1.0:
        return logits
    srt, idx = torch.sort(logits, dim=-1, descending=True)
    cum = srt.softmax(-1).cumsum(-1)
    keep = cum <= top_p
    keep[..., 0] = True
    out = torch.full_like(logits, float("-inf"))
    return out.scatter(-1, idx, torch.where(keep, srt, torch.full_like(srt, float("-inf"))))


def gibbs_step_ref(x, seq, logits, temperature, top_p, n_unmask, u, vocab: int = NVALID, invalid_ids=(), pos_key=None):
    """x, seq (B,L) int64; logits (B,L,>=vocab); u (B,L,4096) uniforms.  Returns new x, entropy, sampled.
    esm's order of operations: entropy and top-p on the WHOLE structure-logit row (`vocab` = 4096
    for the stock head, 4101 for the ESMDiff head), the special ids >= 4096 are masked AFTER top-p, then temperature.
    (A row whose nucleus holds special ids only has no valid candidate left; the best valid id is taken.)"""
    z = logits[..., :vocab].float()
    logp = torch.log_softmax(z, -1)
    ent = -(logp.exp() * logp).sum(-1)
    zp = top_p_logits(z, top_p)[..., :NVALID].clone()    # mask invalid ids after top-p: the specials ...
    zv = z[..., :NVALID].clone()
    for v in invalid_ids:                                # ... and GenerationConfig.invalid_ids, the same way
        zp[..., int(v)] = float("-inf")
        zv[..., int(v)] = float("-inf")
    dead = torch.isinf(zp).all(-1, keepdim=True)
    best_valid = torch.nn.functional.one_hot(zv.argmax(-1), NVALID).bool()
    zp = torch.where(dead & best_valid, torch.zeros_like(zp), zp)
    if temperature == 0:                                  # esm's sample_logits: arg-max of the filtered logits, no noise
        sampled = zp.argmax(-1)
    else:
        w = torch.softmax(zp / temperature, -1)
        g = 1e-10 - (u + 1e-10).log()
        sampled = (w / g).argmax(-1)
    x = x.clone()
    for b in range(x.shape[0]):
        elig = (x[b] == MASK) & (seq[b] != 0) & (seq[b] != 1) & (seq[b] != 2)
        k = int(n_unmask[b])
        if k <= 0 or not bool(elig.any()):
            continue
        # strategy "entropy": lowest entropy first; "random" (pos_key given): lowest random key first = a uniform k-subset
        key = ent[b] if pos_key is None else torch.as_tensor(pos_key[b], dtype=torch.float32)
        e = torch.where(elig, key, torch.full_like(key, float("inf")))
        order = sorted(range(x.shape[1]), key=lambda i: (float(e[i]), i))[:k]
        for i in order:
            if bool(elig[i]):
                x[b, i] = sampled[b, i]
    return x, ent, sampled
