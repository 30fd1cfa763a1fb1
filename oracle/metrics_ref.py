"""TEST INFRASTRUCTURE — numpy restatement of the reference's ensemble metrics on CA traces
(/root/reference/slm/utils/eval_utils.py: pairwise_distance_ca :90-102, radius_of_gyration :105-129, _steric_clash
:132-155, validity :158-173, bonding_validity :176-188, js_pwd :227-255, js_rg :290-316), written out step by step
(numpy.histogram's equal-width binning and scipy's jensenshannon included) the way csrc/metrics.hip computes them.
PINNED: reproduces tests/golden/g9_metrics.npz, which the reference's own functions produced
(tests/golden/make_goldens_metrics.py).  Values are returned UNROUNDED; the reference rounds to 4 decimals last."""
from __future__ import annotations

import numpy as np

PSEUDO_C = 1e-6


def pairwise_distance_ca(coords: np.ndarray, k: int = 1) -> np.ndarray:
    L = coords.shape[-2]
    row, col = np.triu_indices(L, k=k)
    d = coords[..., col, :] - coords[..., row, :]
    return np.sqrt((d[..., 0] ** 2 + d[..., 1] ** 2) + d[..., 2] ** 2)


def radius_of_gyration(coords: np.ndarray) -> np.ndarray:
    L = coords.shape[-2]
    centered = coords - coords.mean(-2, keepdims=True)
    return ((centered ** 2).sum(-1) * (np.ones(L) / L)).sum(-1) ** 0.5


def histogram_equal_width(x: np.ndarray, n_bins: int, first: float, last: float) -> np.ndarray:
    """numpy.histogram(x, bins=n_bins, range=(first, last))[0] for unit weights, as numpy 2.x computes it."""
    if first == last:
        first, last = first - 0.5, last + 0.5
    edges = np.arange(0, n_bins + 1) * ((last - first) / n_bins) + first
    edges[-1] = last
    x = x[(x >= first) & (x <= last)]
    idx = (((x - first) / (last - first)) * n_bins).astype(np.intp)
    idx[idx == n_bins] -= 1
    idx[x < edges[idx]] -= 1
    inc = (x >= edges[idx + 1]) & (idx != n_bins - 1)
    idx[inc] += 1
    return np.bincount(idx, minlength=n_bins).astype(np.float64)


def jensenshannon(p: np.ndarray, q: np.ndarray) -> float:
    p, q = p / p.sum(), q / q.sum()
    m = (p + q) / 2.0
    left = np.where(p > 0, p * np.log(p / m), 0.0)
    right = np.where(q > 0, q * np.log(q / m), 0.0)
    return float(np.sqrt((left.sum() + right.sum()) / 2.0))


def js_columns(model: np.ndarray, ref: np.ndarray, n_bins: int) -> float:
    """mean over columns of JS(hist(model[:, d]), hist(ref[:, d])), bins spanning the reference's [min, max] per column."""
    out = []
    for d in range(ref.shape[1]):
        lo, hi = ref[:, d].min(), ref[:, d].max()
        out.append(jensenshannon(histogram_equal_width(model[:, d], n_bins, lo, hi) + PSEUDO_C,
                                 histogram_equal_width(ref[:, d], n_bins, lo, hi) + PSEUDO_C))
    return float(np.mean(out))


def js_pwd(model_ca, ref_ca, n_bins=50, pwd_offset=3) -> float:
    return js_columns(pairwise_distance_ca(model_ca, pwd_offset), pairwise_distance_ca(ref_ca, pwd_offset), n_bins)


def js_rg(model_ca, ref_ca, n_bins=50) -> float:
    return js_columns(radius_of_gyration(model_ca)[:, None], radius_of_gyration(ref_ca)[:, None], n_bins)


def validity(ca, ca_vdw_radius=1.7, allowable_overlap=0.4, k_exclusion=0) -> float:
    bar = 2 * ca_vdw_radius - allowable_overlap
    n_clash = (pairwise_distance_ca(ca, k_exclusion + 1) < bar).sum(-1)
    return float(1.0 - (n_clash > 0).mean())


def bonding_validity(model_ca, ref_ca) -> float:
    thres = pairwise_adjacent(ref_ca).max() + 1e-6
    return float((pairwise_adjacent(model_ca) < thres).all(-1).sum() / len(model_ca))


def pairwise_adjacent(coords):
    d = coords[..., :-1, :] - coords[..., 1:, :]
    return np.sqrt((d[..., 0] ** 2 + d[..., 1] ** 2) + d[..., 2] ** 2)
