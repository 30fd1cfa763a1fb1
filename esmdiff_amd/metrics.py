"""Ensemble metrics on the device, with the reference's call shapes
(/root/reference/slm/utils/eval_utils.py: js_pwd :227-255, js_rg :290-316, validity :158-173, bonding_validity :176-188):
dictionaries {name: CA coordinates (n_frames, L, 3)} in, dictionaries {name: value rounded to 4 decimals} out, the entry
`ref_key` being the reference ensemble.  The arithmetic runs in csrc/metrics.hip (float64); there is no CPU fallback.
Not covered: js_tica (needs deeptime's TICA fit), per-frame weights, the kl=True variants."""
from __future__ import annotations

import ctypes
from typing import Dict

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


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _call(fn, *args) -> float:
    out = ctypes.c_double(0.0)
    code = fn(*args, ctypes.byref(out), _stream())
    if code != 0:
        raise RuntimeError(f"libesmdiff_hip metrics call failed ({code})")
    return float(out.value)


def js_pwd(ca_coords_dict: Dict[str, np.ndarray], ref_key: str = "target", n_bins: int = 50, pwd_offset: int = 3,
           rounded: bool = True) -> Dict[str, float]:
    L_ = N.lib()
    dev = {k: _dev(v) for k, v in ca_coords_dict.items()}
    ref = dev[ref_key]
    res = {k: _call(L_.esmdiff_metrics_js_pwd, v.data_ptr(), v.shape[0], ref.data_ptr(), ref.shape[0], ref.shape[1], n_bins,
                    pwd_offset) for k, v in dev.items() if k != ref_key}
    res[ref_key] = 0.0
    return {k: float(np.around(v, decimals=4)) if rounded else v for k, v in res.items()}


def js_rg(ca_coords_dict, ref_key: str = "target", n_bins: int = 50, rounded: bool = True) -> Dict[str, float]:
    L_ = N.lib()
    dev = {k: _dev(v) for k, v in ca_coords_dict.items()}
    ref = dev[ref_key]
    res = {k: _call(L_.esmdiff_metrics_js_rg, v.data_ptr(), v.shape[0], ref.data_ptr(), ref.shape[0], ref.shape[1], n_bins)
           for k, v in dev.items() if k != ref_key}
    res[ref_key] = 0.0
    return {k: float(np.around(v, decimals=4)) if rounded else v for k, v in res.items()}


def validity(ca_coords_dict, ca_vdw_radius: float = 1.7, allowable_overlap: float = 0.4, k_exclusion: int = 0,
             rounded: bool = True) -> Dict[str, float]:
    L_ = N.lib()
    res = {}
    for k, v in ca_coords_dict.items():
        d = _dev(v)
        assert not bool(torch.isnan(d).any()), "coords should not contain nan"
        res[k] = _call(L_.esmdiff_metrics_validity, d.data_ptr(), d.shape[0], d.shape[1], float(ca_vdw_radius),
                       float(allowable_overlap), int(k_exclusion))
    return {k: float(np.around(v, decimals=4)) if rounded else v for k, v in res.items()}


def bonding_validity(ca_coords_dict, ref_key: str = "target", rounded: bool = True) -> Dict[str, float]:
    L_ = N.lib()
    dev = {k: _dev(v) for k, v in ca_coords_dict.items()}
    ref = dev[ref_key]
    res = {k: _call(L_.esmdiff_metrics_bonding_validity, v.data_ptr(), v.shape[0], ref.data_ptr(), ref.shape[0], ref.shape[1])
           for k, v in dev.items()}
    return {k: float(np.around(v, decimals=4)) if rounded else v for k, v in res.items()}
