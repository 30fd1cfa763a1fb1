"""Ensemble metrics on the device, with the reference's call shapes
(/root/reference/slm/utils/eval_utils.py: js_pwd :227-255, js_tica :258-289, js_rg :290-316, validity :158-173,
bonding_validity :176-188): dictionaries {name: CA coordinates (n_frames, L, 3)} in, dictionaries {name: value rounded to
4 decimals} out, the entry `ref_key` being the reference ensemble; per-frame `weights=` dictionaries and the `kl=True`
variants included.  The arithmetic runs in csrc/metrics.hip (float64); there is no CPU fallback.

js_tica: the reference fits deeptime's TICA(dim=2, lagtime=20) on the reference ensemble's pairwise distances.  deeptime is
not in the reference tree and not installable here, so the fit is restated ([DEEPTIME-RECALL], PARITY UNPINNED): reversible
(symmetrised) instantaneous / time-lagged covariances of the mean-free features, rank truncation of the instantaneous
covariance at `epsilon = 1e-6`, symmetric eigenproblem in the whitened space, the two leading eigenvectors.  The Jensen-Shannon
value only depends on the two TIC *directions*: the histogram range follows the reference projection, so sign and
kinetic-map scaling cancel.  Dense f64 linear algebra (covariances, eigh) goes through torch on the GPU (rocBLAS / rocSOLVER:
plain library calls, off the hot path); features, histograms and JS are the kernels of csrc/metrics.hip."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import _native as N


def _dev(ca) -> torch.Tensor:
    if not torch.cuda.is_available():
        raise RuntimeError("esmdiff_amd.metrics needs an MI355X (gfx950); there is no CPU fallback")
    t = torch.as_tensor(np.asarray(ca) if not torch.is_tensor(ca) else ca)
    if t.dim() != 3 or t.shape[-1] != 3:
        raise AssertionError(f"CA coords should be 3D (n_frames, L, 3), got {tuple(t.shape)}")
    return t.to(device="cuda", dtype=torch.float64).contiguous()


def _wdev(weights: Optional[dict], key, n: int) -> Optional[torch.Tensor]:
    """The reference fills missing keys with ones (eval_utils.py:236-238); None keeps the unit-weight fast path."""
    if weights is None or key not in weights:
        return None
    w = torch.as_tensor(np.asarray(weights[key], dtype=np.float64)).to("cuda").contiguous()
    assert w.shape == (n,), f"weights[{key!r}] must have one entry per frame ({n}), got {tuple(w.shape)}"
    return w


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _call(fn, *args) -> float:
    out = ctypes.c_double(0.0)
    code = fn(*args, ctypes.byref(out), _stream())
    if code != 0:
        raise RuntimeError(f"libesmdiff_hip metrics call failed ({code})")
    return float(out.value)


def _round(res: dict, rounded: bool) -> dict:
    return {k: float(np.around(v, decimals=4)) if rounded else v for k, v in res.items()}


def js_pwd(ca_coords_dict: Dict[str, np.ndarray], ref_key: str = "target", n_bins: int = 50, pwd_offset: int = 3,
           weights: Optional[dict] = None, kl: bool = False, rounded: bool = True) -> Dict[str, float]:
    L_ = N.lib()
    dev = {k: _dev(v) for k, v in ca_coords_dict.items()}
    ref = dev[ref_key]
    wr = _wdev(weights, ref_key, ref.shape[0])
    res = {k: _call(L_.esmdiff_metrics_js_pwd, _p(v), v.shape[0], _p(_wdev(weights, k, v.shape[0])), _p(ref), ref.shape[0],
                    _p(wr), ref.shape[1], n_bins, pwd_offset, int(kl)) for k, v in dev.items() if k != ref_key}
    res[ref_key] = 0.0
    return _round(res, rounded)


def js_rg(ca_coords_dict, ref_key: str = "target", n_bins: int = 50, weights: Optional[dict] = None, kl: bool = False,
          rounded: bool = True) -> Dict[str, float]:
    L_ = N.lib()
    dev = {k: _dev(v) for k, v in ca_coords_dict.items()}
    ref = dev[ref_key]
    wr = _wdev(weights, ref_key, ref.shape[0])
    res = {k: _call(L_.esmdiff_metrics_js_rg, _p(v), v.shape[0], _p(_wdev(weights, k, v.shape[0])), _p(ref), ref.shape[0],
                    _p(wr), ref.shape[1], n_bins, int(kl)) for k, v in dev.items() if k != ref_key}
    res[ref_key] = 0.0
    return _round(res, rounded)


def pairwise_distance_ca(ca, k: int = 1) -> torch.Tensor:
    """eval_utils.py:90-102 on the device: (n, L, 3) -> (n, D) CA distances of the pairs (i, j >= i + k), triu order."""
    d = _dev(ca)
    n, L = d.shape[:2]
    D = (L - k) * (L - k + 1) // 2
    out = torch.empty(n, D, dtype=torch.float64, device="cuda")
    code = N.lib().esmdiff_metrics_pwd(_p(d), n, L, k, _p(out), _stream())
    if code != 0:
        raise RuntimeError(f"libesmdiff_hip metrics call failed ({code})")
    return out


def tica_fit(x: torch.Tensor, lagtime: int, dim: int = 2, epsilon: float = 1e-6):
    """[DEEPTIME-RECALL] TICA(dim, lagtime).fit(x) restated (module docstring): -> (mean (D,), directions (D, dim))."""
    n = x.shape[0]
    if n <= lagtime:
        raise ValueError(f"TICA needs more than lagtime = {lagtime} frames in the reference ensemble, got {n}")
    x0, xt = x[:-lagtime], x[lagtime:]
    mean = 0.5 * (x0.mean(0) + xt.mean(0))                       # reversible estimator: one mean for both windows
    a, b = x0 - mean, xt - mean
    m = a.shape[0]
    c00 = (a.T @ a + b.T @ b) / (2.0 * m)
    c0t = (a.T @ b + b.T @ a) / (2.0 * m)
    s, u = torch.linalg.eigh(c00)
    keep = s > epsilon                                           # rank truncation of the instantaneous covariance
    if int(keep.sum()) < dim:
        raise ValueError("reference ensemble has fewer than `dim` directions above the TICA rank cut-off")
    wh = u[:, keep] / torch.sqrt(s[keep])                        # whitening: wh^T c00 wh = I
    lam, v = torch.linalg.eigh(wh.T @ c0t @ wh)
    order = torch.argsort(lam.abs(), descending=True)[:dim]
    # deeptime's TICA defaults to scaling="kinetic_map": the projected coordinates are the whitened eigenfunctions TIMES
    # their eigenvalues.  The JS value does not depend on it (the histogram range follows the reference
    # projection), the returned TIC coordinates do.  The sign of each TIC is arbitrary (an eigenvector's sign).
    return mean, (wh @ v[:, order]) * lam[order]


def js_tica(ca_coords_dict, ref_key: str = "target", n_bins: int = 50, lagtime: int = 20, return_tic: bool = True,
            weights: Optional[dict] = None, rounded: bool = True):
    """eval_utils.py:258-289: coordinates -> pairwise distances (k = 1) -> two TICs fitted on the reference -> JS."""
    L_ = N.lib()
    pwd = {k: pairwise_distance_ca(v) for k, v in ca_coords_dict.items()}
    mean, comp = tica_fit(pwd[ref_key], lagtime)
    dr = {k: ((v - mean) @ comp).contiguous() for k, v in pwd.items()}
    ref = dr[ref_key]
    wr = _wdev(weights, ref_key, ref.shape[0])
    res = {k: _call(L_.esmdiff_metrics_js_columns, _p(v), v.shape[0], _p(_wdev(weights, k, v.shape[0])), _p(ref), ref.shape[0],
                    _p(wr), 2, n_bins, 0) for k, v in dr.items() if k != ref_key}
    res[ref_key] = 0.0
    res = _round(res, rounded)
    if return_tic:
        return res, {k: v.cpu().numpy() for k, v in dr.items()}
    return res


def validity(ca_coords_dict, ca_vdw_radius: float = 1.7, allowable_overlap: float = 0.4, k_exclusion: int = 0,
             rounded: bool = True) -> Dict[str, float]:
    L_ = N.lib()
    res = {}
    for k, v in ca_coords_dict.items():
        d = _dev(v)
        assert not bool(torch.isnan(d).any()), "coords should not contain nan"
        res[k] = _call(L_.esmdiff_metrics_validity, d.data_ptr(), d.shape[0], d.shape[1], float(ca_vdw_radius),
                       float(allowable_overlap), int(k_exclusion))
    return _round(res, rounded)


def bonding_validity(ca_coords_dict, ref_key: str = "target", rounded: bool = True) -> Dict[str, float]:
    L_ = N.lib()
    dev = {k: _dev(v) for k, v in ca_coords_dict.items()}
    ref = dev[ref_key]
    res = {k: _call(L_.esmdiff_metrics_bonding_validity, v.data_ptr(), v.shape[0], ref.data_ptr(), ref.shape[0], ref.shape[1])
           for k, v in dev.items()}
    return _round(res, rounded)


# ---- the small helpers around the metrics (eval_utils.py:37-129, :191-224), on the device, float64 ---------------------------
def radius_of_gyration(coords, masses=None) -> torch.Tensor:
    """eval_utils.py:105-129: (n, L, 3) -> (n,).  weights = masses / sum (equal masses when None); the centre is the plain mean."""
    d = _dev(coords)
    L = d.shape[1]
    if masses is None:
        m = torch.ones(L, dtype=torch.float64, device="cuda")
    else:
        m = torch.as_tensor(np.asarray(masses, dtype=np.float64)).to("cuda")
        assert m.dim() == 1, f"masses should be 1D, got {tuple(m.shape)}"
        assert m.shape[0] == L, f"masses {tuple(m.shape)} != number of particles {L}"
    w = m / m.sum()
    centered = d - d.mean(-2, keepdim=True)
    return (((centered ** 2).sum(-1) * w).sum(-1)) ** 0.5


def rmsf(coords) -> torch.Tensor:
    """eval_utils.py:51-54: (n, L, 3) -> (L,)."""
    d = _dev(coords)
    return torch.sqrt(torch.var(d, dim=0, unbiased=False).mean(-1))


def adjacent_ca_distance(coords) -> torch.Tensor:
    """eval_utils.py:64-74: (n, L, 3) -> (n, L - 1)."""
    d = _dev(coords)
    dx = d[:, :-1] - d[:, 1:]
    return torch.sqrt((dx ** 2).sum(-1))


def distance_matrix_ca(coords) -> torch.Tensor:
    """eval_utils.py:77-87: (n, L, 3) -> (n, L, L)."""
    d = _dev(coords)
    dx = d[:, None, :, :] - d[:, :, None, :]
    return torch.sqrt((dx ** 2).sum(-1))


def idp_metrics(ca_coords_dict, ref_key: str = "target", pwd_offset: int = 3):
    """eval_utils.py:191-224: (mse_pwd, mse_rg, mse_contact, mae_pwd, mae_rg, mae_contact) dictionaries — mean pairwise distances
    (csrc/metrics.hip's pair kernel), mean radius of gyration and log contact probabilities (< 8 A, pseudo-count 0.01) of every
    ensemble against the reference ensemble's."""
    pseudo_c = 0.01

    def stats(ca):
        pwd = pairwise_distance_ca(ca, k=pwd_offset)
        return pwd.mean(0), radius_of_gyration(ca).mean(0), torch.log((pwd < 8.0).to(torch.float64).mean(0) + pseudo_c)

    ref = stats(ca_coords_dict[ref_key])
    out = tuple({} for _ in range(6))
    for name, ca in ca_coords_dict.items():
        cur = stats(ca)
        for j in range(3):
            diff = cur[j] - ref[j]
            out[j][name] = float((diff ** 2).mean())
            out[3 + j][name] = float(diff.abs().mean())
    return out


def position_specific_entropy(tokens: torch.Tensor) -> torch.Tensor:
    """eval_utils.py:37-49: tokens (n_frames, L) int64 -> (L,) float32 entropy in bits of each column's token frequencies; one
    scatter-add over a (L, n_ids) count table instead of a Python loop over columns."""
    t = torch.as_tensor(tokens).to(device="cuda", dtype=torch.int64)
    n, L = t.shape
    n_ids = int(t.max()) + 1
    counts = torch.zeros(L, n_ids, dtype=torch.float32, device="cuda")
    counts.scatter_add_(1, t.t().contiguous(), torch.ones(L, n, dtype=torch.float32, device="cuda"))
    f = counts / n
    return -(torch.where(f > 0, f * torch.log2(f.clamp_min(1e-38)), torch.zeros_like(f))).sum(-1)
