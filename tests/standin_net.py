"""Tiny deterministic stand-in for the ESM3 network, used ONLY by tests and by
tests/golden/make_goldens.py.

It exposes the keyword signature of the reference's CustomizedESM3.forward
(/root/reference/slm/models/net.py:371-389) and returns an object with a
`.structure_logits` attribute, so that the reference's (and the oracle's)
`_model_wrapper` can drive it.  Its arithmetic is table look-ups plus
elementwise float32 adds/multiplies only (no matmul, no reduction), so the
logits are bit-reproducible on any machine; the weights come from numpy's
frozen legacy RandomState and therefore need not be committed.
"""
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

V = 4101


def _rs(seed):
    return np.random.RandomState(seed)


def standin_sigma_embedder_state(hidden, freq=256):
    """State dict for TimestepEmbedder(hidden) (net.py:486-522 key layout mlp.{0,2}.{weight,bias})."""
    r = _rs(1000 + hidden)
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))
    return {
        "mlp.0.weight": f32(r.standard_normal((hidden, freq)) / np.sqrt(freq)),
        "mlp.0.bias": f32(r.standard_normal((hidden,)) * 0.1),
        "mlp.2.weight": f32(r.standard_normal((hidden, hidden)) / np.sqrt(hidden)),
        "mlp.2.bias": f32(r.standard_normal((hidden,)) * 0.1),
    }


class StandinNet(nn.Module):
    P_X, P_S = 13, 7

    def __init__(self, aux_dim):
        super().__init__()
        r = _rs(77)
        f32 = lambda a: torch.from_numpy(a.astype(np.float32))
        self.register_buffer("tab_x", f32(r.standard_normal((self.P_X, V)) * 2.0))
        self.register_buffer("tab_s", f32(r.standard_normal((self.P_S, V)) * 1.5))
        self.register_buffer("tab_a", f32(r.standard_normal((V,)) * 0.5))
        self.register_buffer("tab_p", f32(r.standard_normal((64, V)) * 1.0))
        self.aux_dim = aux_dim

    def forward(self, structure_tokens=None, sequence_tokens=None, auxiliary_embeddings=None,
                labels=None, **kw):
        x = structure_tokens
        B, L = x.shape
        logits = self.tab_x[x % self.P_X] + self.tab_s[sequence_tokens % self.P_S]
        logits = logits + self.tab_p[torch.arange(L) % 64][None]
        if auxiliary_embeddings is not None:
            a = auxiliary_embeddings[..., 0:1] * 0.25 + auxiliary_embeddings[..., 1:2] * 0.125
            logits = logits + a * self.tab_a
        return SimpleNamespace(structure_logits=logits.contiguous())
