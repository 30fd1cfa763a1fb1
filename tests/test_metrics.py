"""Ensemble metrics: the reference's numpy/scipy functions (eval_utils.py:90-316) on the device.

g9_metrics.npz holds inputs and outputs produced by the reference's own functions (tests/golden/make_goldens_metrics.py,
values recorded before and after its 4-decimal rounding).  CPU: the numpy restatement in oracle/metrics_ref.py reproduces
them.  GPU: csrc/metrics.hip through the C ABI reproduces them (floating point: |err| <= 1e-12 on the unrounded values,
identical after rounding) and follows the oracle on other sizes, including degenerate columns and values outside the
reference's range."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import metrics_ref as R

G = np.load(Path(__file__).parent / "golden" / "g9_metrics.npz")
KEYS = [str(k) for k in G["keys"]]
ENS = {k: G["ca_" + k] for k in KEYS}
TOL = 1e-12


def test_g9_oracle_reproduces_reference_metrics():
    assert np.array_equal(R.pairwise_distance_ca(ENS["target"], 3), G["pwd_target_k3"])     # bit-exact distances
    assert float(np.abs(R.radius_of_gyration(ENS["target"]) - G["rg_target"]).max()) < 1e-14
    for i, k in enumerate(KEYS):
        if k != "target":
            assert abs(R.js_pwd(ENS[k], ENS["target"]) - G["js_pwd_raw"][i]) < TOL
            assert abs(R.js_pwd(ENS[k], ENS["target"], 20, 1) - G["js_pwd_k1_b20_raw"][i]) < TOL
            assert abs(R.js_rg(ENS[k], ENS["target"]) - G["js_rg_raw"][i]) < TOL
        assert abs(R.validity(ENS[k]) - G["validity_raw"][i]) < TOL
        assert abs(R.bonding_validity(ENS[k], ENS["target"]) - G["bonding_validity_raw"][i]) < TOL
    assert 0 < G["validity_raw"][1] < 1 and 0 < G["bonding_validity_raw"][2] < 1             # the fixture is not degenerate


def test_histogram_matches_numpy_on_edges():
    rng = np.random.default_rng(0)
    for n_bins in (1, 7, 50):
        lo, hi = -1.25, 3.5
        edges = np.linspace(lo, hi, n_bins + 1)
        x = np.concatenate([rng.uniform(lo - 1, hi + 1, 500), edges, np.nextafter(edges, 10), np.nextafter(edges, -10)])
        assert np.array_equal(R.histogram_equal_width(x, n_bins, lo, hi), np.histogram(x, bins=n_bins, range=(lo, hi))[0])
    x = np.full(9, 2.0)
    assert np.array_equal(R.histogram_equal_width(x, 5, 2.0, 2.0), np.histogram(x, bins=5, range=(2.0, 2.0))[0])


@pytest.mark.gpu
def test_g9_device_metrics_reproduce_reference():
    from esmdiff_amd import metrics as M
    raw = {"js_pwd": M.js_pwd(ENS, rounded=False), "js_pwd_k1_b20": M.js_pwd(ENS, n_bins=20, pwd_offset=1, rounded=False),
           "js_rg": M.js_rg(ENS, rounded=False), "validity": M.validity(ENS, rounded=False),
           "bonding_validity": M.bonding_validity(ENS, rounded=False)}
    rnd = {"js_pwd": M.js_pwd(ENS), "js_pwd_k1_b20": M.js_pwd(ENS, n_bins=20, pwd_offset=1), "js_rg": M.js_rg(ENS),
           "validity": M.validity(ENS), "bonding_validity": M.bonding_validity(ENS)}
    for name in raw:
        for i, k in enumerate(KEYS):
            assert abs(raw[name][k] - G[name + "_raw"][i]) < TOL, (name, k, raw[name][k], G[name + "_raw"][i])
            assert rnd[name][k] == float(G[name + "_rounded"][i]), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("L,n_ref,n_model", [(5, 3, 2), (60, 200, 100), (258, 100, 100)])
def test_device_metrics_follow_the_oracle(L, n_ref, n_model):
    from esmdiff_amd import metrics as M
    rng = np.random.default_rng(L)
    steps = rng.normal(size=(L, 3)) + np.array([2.0, 0, 0])
    base = np.cumsum(3.8 * steps / np.linalg.norm(steps, axis=-1, keepdims=True), 0)
    ref = base[None] + rng.normal(size=(n_ref, L, 3)) * 0.2
    mod = base[None] * (1 + 0.05 * rng.normal(size=(n_model, 1, 1))) + rng.normal(size=(n_model, L, 3)) * 0.4
    ref[:, 1] = ref[0, 1]                                # constant columns: degenerate histogram range (lo == hi)
    ref[:, 2] = ref[0, 2]
    ens = {"target": ref, "m": mod}
    k = min(3, L - 1)
    assert abs(M.js_pwd(ens, pwd_offset=k, rounded=False)["m"] - R.js_pwd(mod, ref, 50, k)) < TOL
    assert abs(M.js_rg(ens, rounded=False)["m"] - R.js_rg(mod, ref)) < TOL
    for name in ens:
        assert abs(M.validity(ens, rounded=False)[name] - R.validity(ens[name])) < TOL
        assert abs(M.validity(ens, k_exclusion=2, rounded=False)[name] - R.validity(ens[name], k_exclusion=2)) < TOL
        assert abs(M.bonding_validity(ens, rounded=False)[name] - R.bonding_validity(ens[name], ref)) < TOL
    with pytest.raises(AssertionError):
        M.js_pwd({"target": ref[0], "m": mod})
