#!/usr/bin/env python3
"""Golden vectors for the ensemble metrics, produced by the REFERENCE's own code:
/root/reference/slm/utils/eval_utils.py  js_pwd :227-255, js_rg :290-316, validity :158-173,
bonding_validity :176-188, pairwise_distance_ca :90-102, radius_of_gyration :105-129.

Runs only in the build container (needs /root/reference); importing make_goldens installs the stub modules for the
packages eval_utils drags in (deeptime, slm.utils.protein's Bio, ...).  To record the values BEFORE the reference rounds
them to four decimals, numpy.around is replaced by the identity while the reference functions run; the rounded
results are recorded too.  Fixture: tests/golden/g9_metrics.npz.
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_goldens  # noqa: F401,E402  (installs the stubs and puts /root/reference on sys.path)
from slm.utils import eval_utils as E  # noqa: E402


def template(rng, L):
    """One CA trace: 3.8 A steps with a drift, i.e. an extended chain without self-contacts."""
    steps = rng.normal(size=(L, 3)) + np.array([2.0, 0.0, 0.0])
    steps /= np.linalg.norm(steps, axis=-1, keepdims=True)
    return np.cumsum(steps * 3.8, axis=0)


def ensemble(rng, base, n_frames, spread, breathe):
    """Frames of the same protein: per-frame isotropic scaling about the centroid (moves Rg) + per-atom noise."""
    c = base.mean(0, keepdims=True)
    scale = 1.0 + breathe * rng.normal(size=(n_frames, 1, 1))
    frames = c[None] + (base - c)[None] * scale + rng.normal(size=(n_frames,) + base.shape) * spread
    return np.ascontiguousarray(frames, dtype=np.float64)


def main():
    rng = np.random.default_rng(20250928)
    L = 24
    base = template(rng, L)
    ens = {"target": ensemble(rng, base, 37, 0.10, 0.03), "model_a": ensemble(rng, base, 25, 0.25, 0.05),
           "model_b": ensemble(rng, base, 11, 0.05, 0.01)}
    ens["model_b"][3, 7] = ens["model_b"][3, 10] + 0.2           # one steric clash
    ens["model_a"][5, 12:] += 2.5                                  # one stretched bond
    out = {f"ca_{k}": v for k, v in ens.items()}
    out["pwd_target_k3"] = E.pairwise_distance_ca(ens["target"], k=3)
    out["rg_target"] = E.radius_of_gyration(ens["target"])
    funcs = {"js_pwd": lambda: E.js_pwd(dict(ens)), "js_pwd_k1_b20": lambda: E.js_pwd(dict(ens), n_bins=20, pwd_offset=1),
             "js_rg": lambda: E.js_rg(dict(ens)), "validity": lambda: E.validity(dict(ens)),
             "bonding_validity": lambda: E.bonding_validity(dict(ens))}
    keys = list(ens)
    for name, fn in funcs.items():
        rounded = fn()
        keep = np.around
        np.around = lambda v, decimals=0: v
        try:
            raw = fn()
        finally:
            np.around = keep
        out[name + "_rounded"] = np.array([float(rounded[k]) for k in keys])
        out[name + "_raw"] = np.array([float(raw[k]) for k in keys])
    out["keys"] = np.array(keys)
    np.savez_compressed(HERE / "g9_metrics.npz", **out)
    for name in funcs:
        print(name, dict(zip(keys, out[name + "_raw"])))

    # g9b (r02): the weights= and kl=True arguments of js_pwd / js_rg (eval_utils.py:227, :290), same ensembles
    w = {"target": rng.uniform(0.2, 2.0, size=len(ens["target"])), "model_a": rng.uniform(0.2, 2.0, size=len(ens["model_a"]))}
    outb = {"w_target": w["target"], "w_model_a": w["model_a"]}          # model_b: no entry -> the reference fills in ones
    funcs_b = {"js_pwd_w": lambda: E.js_pwd(dict(ens), weights=dict(w)), "js_rg_w": lambda: E.js_rg(dict(ens), weights=dict(w)),
               "js_pwd_kl": lambda: E.js_pwd(dict(ens), kl=True), "js_rg_kl": lambda: E.js_rg(dict(ens), kl=True),
               "js_pwd_w_kl_b20": lambda: E.js_pwd(dict(ens), n_bins=20, weights=dict(w), kl=True)}
    for name, fn in funcs_b.items():
        keep = np.around
        np.around = lambda v, decimals=0: v
        try:
            raw = fn()
        finally:
            np.around = keep
        outb[name + "_raw"] = np.array([float(raw[k]) for k in keys])
        print(name, dict(zip(keys, outb[name + "_raw"])))
    outb["keys"] = np.array(keys)
    np.savez_compressed(HERE / "g9b_metrics_weights.npz", **outb)
    g9c(ens, keys, rng)


def g9c(ens, keys, rng):
    """g9c (r04): the small ensemble helpers around the metrics — idp_metrics :191-224, rmsf :51-54, adjacent_ca_distance :64-74,
    distance_matrix_ca :77-87, radius_of_gyration with masses :105-129, position_specific_entropy :37-49, split_pdbfile :495-530
    (on the merged file of g8)."""
    import json
    import tempfile

    import torch
    out = {}
    names = ("mse_pwd", "mse_rg", "mse_contact", "mae_pwd", "mae_rg", "mae_contact")
    for tag, kw in (("", {}), ("_k1", {"pwd_offset": 1})):
        for nm, d in zip(names, E.idp_metrics(dict(ens), **kw)):
            out[f"idp_{nm}{tag}"] = np.array([float(d[k]) for k in keys])
    out["rmsf_target"] = E.rmsf(ens["target"])
    out["adjacent_model_a"] = E.adjacent_ca_distance(ens["model_a"])
    out["distance_matrix_model_b"] = E.distance_matrix_ca(ens["model_b"])
    masses = rng.uniform(10.0, 20.0, size=ens["target"].shape[1])
    out["masses"] = masses
    out["rg_target_masses"] = E.radius_of_gyration(ens["target"], masses=masses)
    tok = torch.from_numpy(rng.integers(0, 7, size=(40, 9)))
    tok[:, 2] = 3                                                   # a position with one token only: entropy 0
    out["tokens"] = tok.numpy()
    out["position_specific_entropy"] = E.position_specific_entropy(tok).numpy()
    np.savez_compressed(HERE / "g9c_ensemble_helpers.npz", **out)
    g8 = json.loads((HERE / "g8_merge_pdb.json").read_text())
    with tempfile.TemporaryDirectory() as d:
        src = Path(d) / "merged.pdb"
        src.write_text(g8["merged"])
        parts = E.split_pdbfile(src, output_dir=Path(d) / "split", verbose=False)
        files = {p.name: p.read_text() for p in sorted((Path(d) / "split").iterdir())}
    (HERE / "g9c_split_pdb.json").write_text(json.dumps({"parts": parts, "files": files}, indent=1))
    print("g9c", {k: out[k].shape for k in out})


if __name__ == "__main__":
    main()
