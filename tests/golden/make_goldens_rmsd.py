#!/usr/bin/env python3
"""Golden vectors for the RMSD the decode-parity bar is stated in (north_star: "decoded backbone RMSD within 1e-4 A"),
produced by the REFERENCE's own code: /root/reference/slm/utils/geo_utils.py squared_deviation :58-88 (reduction='rmsd')
and _find_rigid_alignment :91-122 (Kabsch through torch.svd).

Runs only in the build container (needs /root/reference; geo_utils itself imports torch only, the package __init__
needs make_goldens' stubs for hydra etc.).  Fixture: tests/golden/g10_rmsd.npz
— float64 point clouds, their rigidly moved + perturbed copies, and the reference's RMSD / per-point squared deviation.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_goldens  # noqa: F401,E402  (inert stubs for the packages slm.utils' __init__ drags in; /root/reference on sys.path)
from slm.utils import geo_utils as G  # noqa: E402


def random_rotation(rng):
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def main():
    rng = np.random.default_rng(20250929)
    B, L = 5, 48
    tgt = np.cumsum(rng.normal(size=(B, L, 3)) * 2.2, axis=1)
    src = np.empty_like(tgt)
    noise = [0.0, 1e-5, 1e-3, 0.3, 2.0]
    for b in range(B):
        R, t = random_rotation(rng), rng.normal(size=3) * 20
        src[b] = (tgt[b] + rng.normal(size=(L, 3)) * noise[b]) @ R.T + t
    s, t_ = torch.as_tensor(src), torch.as_tensor(tgt)
    out = {"src": src, "tgt": tgt, "noise": np.array(noise),
           "rmsd": G.squared_deviation(s, t_, reduction="rmsd").numpy(),
           "sd": G.squared_deviation(s, t_, reduction="none").numpy(),
           # the numpy entry path of the same function (map_to_np)
           "rmsd_np_entry": G.squared_deviation(src, tgt, reduction="rmsd")}
    R, t = G._find_rigid_alignment(s, t_)
    out["R"], out["t"] = R.numpy(), t.numpy()
    np.savez(HERE / "g10_rmsd.npz", **out)
    print({k: v.shape for k, v in out.items()}, out["rmsd"])


if __name__ == "__main__":
    main()
