"""TEST INFRASTRUCTURE — numpy restatement of the reference's ensemble metrics on CA traces
(/root/reference/slm/utils/eval_utils.py: pairwise_distance_ca :90-102, radius_of_gyration :105-129, _steric_clash
:132-155, validity :158-173, bonding_validity :176-188, js_pwd :227-255, js_rg :290-316), written out step by step
(numpy.histogram's equal-width binning and scipy's jensenshannon included) the way csrc/metrics.hip computes them.
PINNED: reproduces tests/golden/g9_metrics.npz, which the reference's own functions produced
(tests/golden/make_goldens_metrics.py) — js_pwd / js_rg incl. weights= and kl=True (g9b), validity, bonding_validity.
js_tica (:258-289) is NOT pinned: deeptime's TICA is absent from the reference tree and from this image; its fit is
restated here and in esmdiff_amd/metrics.py ([DEEPTIME-RECALL]).  Values are returned UNROUNDED; the reference rounds to 4 decimals last."""
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


def histogram_equal_width(x: np.ndarray, n_bins: int, first: float, last: float, weights=None) -> np.ndarray:
    """numpy.histogram(x, bins=n_bins, range=(first, last), weights=weights)[0], as numpy 2.x computes it."""
    if first == last:
        first, last = first - 0.5, last + 0.5
    edges = np.arange(0, n_bins + 1) * ((last - first) / n_bins) + first
    edges[-1] = last
    inside = (x >= first) & (x <= last)
    weights = None if weights is None else np.asarray(weights, dtype=np.float64)[inside]
    x = x[inside]
    idx = (((x - first) / (last - first)) * n_bins).astype(np.intp)
    idx[idx == n_bins] -= 1
    idx[x < edges[idx]] -= 1
    inc = (x >= edges[idx + 1]) & (idx != n_bins - 1)
    idx[inc] += 1
    return np.bincount(idx, weights=weights, minlength=n_bins).astype(np.float64)


def jensenshannon(p: np.ndarray, q: np.ndarray) -> float:
    p, q = p / p.sum(), q / q.sum()
    m = (p + q) / 2.0
    left = np.where(p > 0, p * np.log(p / m), 0.0)
    right = np.where(q > 0, q * np.log(q / m), 0.0)
    return float(np.sqrt((left.sum() + right.sum()) / 2.0))


def kl_div(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """scipy.special.kl_div for positive arguments (the histograms carry a pseudo count)."""
    return (p * np.log(p / q) - p) + q


def js_columns(model: np.ndarray, ref: np.ndarray, n_bins: int, w_model=None, w_ref=None, kl: bool = False) -> float:
    """mean over columns of JS(hist(model[:, d]), hist(ref[:, d])), bins spanning the reference's [min, max] per column;
    kl=True: mean of kl_div over all bins and columns (eval_utils.py:247-249)."""
    out = []
    for d in range(ref.shape[1]):
        lo, hi = ref[:, d].min(), ref[:, d].max()
        hm = histogram_equal_width(model[:, d], n_bins, lo, hi, w_model) + PSEUDO_C
        hr = histogram_equal_width(ref[:, d], n_bins, lo, hi, w_ref) + PSEUDO_C
        out.append(kl_div(hm, hr).sum() if kl else jensenshannon(hm, hr))
    return float(np.sum(out) / (ref.shape[1] * n_bins)) if kl else float(np.mean(out))


def js_pwd(model_ca, ref_ca, n_bins=50, pwd_offset=3, w_model=None, w_ref=None, kl=False) -> float:
    return js_columns(pairwise_distance_ca(model_ca, pwd_offset), pairwise_distance_ca(ref_ca, pwd_offset), n_bins,
                      w_model, w_ref, kl)


def js_rg(model_ca, ref_ca, n_bins=50, w_model=None, w_ref=None, kl=False) -> float:
    return js_columns(radius_of_gyration(model_ca)[:, None], radius_of_gyration(ref_ca)[:, None], n_bins, w_model, w_ref, kl)


def tica_fit(x: np.ndarray, lagtime: int, dim: int = 2, epsilon: float = 1e-6):
    """[DEEPTIME-RECALL, PARITY UNPINNED] deeptime.decomposition.TICA(dim, lagtime).fit(x) as eval_utils.py:266 calls it,
    restated with scipy: reversible covariances, rank cut at epsilon, generalised symmetric eigenproblem."""
    import scipy.linalg
    x0, xt = x[:-lagtime], x[lagtime:]
    mean = 0.5 * (x0.mean(0) + xt.mean(0))
    a, b = x0 - mean, xt - mean
    c00 = (a.T @ a + b.T @ b) / (2.0 * len(a))
    c0t = (a.T @ b + b.T @ a) / (2.0 * len(a))
    s, u = scipy.linalg.eigh(c00)
    keep = s > epsilon
    wh = u[:, keep] / np.sqrt(s[keep])
    lam, v = scipy.linalg.eigh(wh.T @ c0t @ wh)
    order = np.argsort(-np.abs(lam))[:dim]
    return mean, wh @ v[:, order], lam[order]


def js_tica(model_ca, ref_ca, n_bins=50, lagtime=20, w_model=None, w_ref=None) -> float:
    pm, pr = pairwise_distance_ca(model_ca, 1), pairwise_distance_ca(ref_ca, 1)
    mean, comp, _ = tica_fit(pr, lagtime)
    return js_columns((pm - mean) @ comp, (pr - mean) @ comp, n_bins, w_model, w_ref)


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


# ---- the small helpers around the metrics (r04; pinned to tests/golden/g9c_ensemble_helpers.npz) ----------------------------
def radius_of_gyration_masses(coords: np.ndarray, masses: np.ndarray) -> np.ndarray:
    """eval_utils.py:105-129 with masses: weights = m / sum(m); the centre stays the unweighted mean (:126)."""
    w = masses / masses.sum()
    centered = coords - coords.mean(-2, keepdims=True)
    return ((centered ** 2).sum(-1) * w).sum(-1) ** 0.5


def rmsf(coords: np.ndarray) -> np.ndarray:
    """eval_utils.py:51-54: sqrt of the per-atom variance over frames, averaged over x, y, z."""
    return np.sqrt(np.mean(np.var(coords, axis=0), axis=-1))


def distance_matrix_ca(coords: np.ndarray) -> np.ndarray:
    """eval_utils.py:77-87: (..., L, 3) -> (..., L, L)."""
    d = coords[..., None, :, :] - coords[..., None, :]
    return np.sqrt(np.sum(d ** 2, axis=-1))


def idp_metrics(ens: dict, ref_key: str = "target", pwd_offset: int = 3):
    """eval_utils.py:191-224: MSE / MAE between an ensemble and the reference ensemble of the mean pairwise distances, the mean
    radius of gyration and the log contact probabilities (pairs closer than 8 A, pseudo-count 0.01)."""
    pseudo_c = 0.01
    ref_pwd = pairwise_distance_ca(ens[ref_key], pwd_offset)
    ref_mean, ref_rg = ref_pwd.mean(axis=0), radius_of_gyration(ens[ref_key]).mean(axis=0)
    ref_con = np.log((ref_pwd < 8.0).mean(axis=0) + pseudo_c)
    out = [dict() for _ in range(6)]
    for name, ca in ens.items():
        pwd = pairwise_distance_ca(ca, pwd_offset)
        rg = radius_of_gyration(ca).mean(axis=0)
        con = np.log((pwd < 8.0).mean(axis=0) + pseudo_c)
        out[0][name] = np.mean((pwd.mean(axis=0) - ref_mean) ** 2)
        out[1][name] = np.mean((rg - ref_rg) ** 2)
        out[2][name] = np.mean((con - ref_con) ** 2)
        out[3][name] = np.mean(np.abs(pwd.mean(axis=0) - ref_mean))
        out[4][name] = np.mean(np.abs(rg - ref_rg))
        out[5][name] = np.mean(np.abs(con - ref_con))
    return tuple(out)


def position_specific_entropy(tokens: np.ndarray) -> np.ndarray:
    """eval_utils.py:37-49: per column, the entropy (bits) of the token frequencies over the frames; float32 like the reference's
    torch.zeros accumulator."""
    n = tokens.shape[0]
    out = np.zeros(tokens.shape[1], dtype=np.float32)
    for c in range(tokens.shape[1]):
        f = (np.bincount(tokens[:, c]) / np.float32(n)).astype(np.float32)
        f = f[f > 0]
        out[c] = -np.sum(f * np.log2(f), dtype=np.float32)
    return out


def split_pdb_text(text: str):
    """eval_utils.py:495-530 on a string: the ATOM / TER records of every MODEL block, each block closed by 'END'."""
    blocks, records = [], []
    for line in text.splitlines(keepends=True):
        name = line[:6].strip()
        if name == "MODEL":
            records = []
        elif name in ("ATOM", "TER"):
            records.append(line)
        elif name in ("ENDMDL", "END") and records:
            blocks.append("".join(records) + "END\n")
            records = []
    return blocks
