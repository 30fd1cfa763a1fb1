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


# ---------------------------------------------------------------------------------------------------
# r02: weights= / kl=True (pinned to the reference's own outputs, g9b) and js_tica (deeptime absent: unpinned)
GB = np.load(Path(__file__).parent / "golden" / "g9b_metrics_weights.npz")
W = {"target": GB["w_target"], "model_a": GB["w_model_a"]}        # model_b has no entry: the reference fills in ones


def test_g9b_oracle_reproduces_reference_weights_and_kl():
    for i, k in enumerate(KEYS):
        if k == "target":
            continue
        wm = W.get(k)
        assert abs(R.js_pwd(ENS[k], ENS["target"], w_model=wm, w_ref=W["target"]) - GB["js_pwd_w_raw"][i]) < TOL
        assert abs(R.js_rg(ENS[k], ENS["target"], w_model=wm, w_ref=W["target"]) - GB["js_rg_w_raw"][i]) < TOL
        assert abs(R.js_pwd(ENS[k], ENS["target"], kl=True) - GB["js_pwd_kl_raw"][i]) < 1e-11
        assert abs(R.js_rg(ENS[k], ENS["target"], kl=True) - GB["js_rg_kl_raw"][i]) < 1e-11
        assert abs(R.js_pwd(ENS[k], ENS["target"], 20, 3, wm, W["target"], True) - GB["js_pwd_w_kl_b20_raw"][i]) < 1e-11


def _trajectory(rng, L, n, lag_corr=0.97):
    """A slowly breathing, slowly bending chain: two slow collective coordinates + fast noise (what TICA should find)."""
    steps = rng.normal(size=(L, 3)) + np.array([2.0, 0, 0])
    base = np.cumsum(3.8 * steps / np.linalg.norm(steps, axis=-1, keepdims=True), 0)
    c = base.mean(0)
    s = np.zeros((n, 2))
    for t in range(1, n):
        s[t] = lag_corr * s[t - 1] + np.sqrt(1 - lag_corr ** 2) * rng.normal(size=2)
    bend = np.linspace(-1, 1, L)[:, None] ** 2 * np.array([0.0, 0.0, 6.0])
    return c + (base - c)[None] * (1 + 0.06 * s[:, :1, None]) + s[:, 1:, None] * bend[None] + rng.normal(size=(n, L, 3)) * 0.15


def test_tica_restatement_finds_the_slow_coordinates():
    rng = np.random.default_rng(3)
    x = R.pairwise_distance_ca(_trajectory(rng, 14, 600), 1)
    mean, comp, lam = R.tica_fit(x, 20)
    assert comp.shape == (x.shape[1], 2) and lam[0] >= lam[1] > 0.2       # two slow modes survive a lag of 20 frames
    y = (x - mean) @ comp
    for j in range(2):                                                     # unit variance, autocorrelation = eigenvalue
        a, b = y[:-20, j], y[20:, j]
        assert abs(0.5 * (a @ a + b @ b) / len(a) - 1.0) < 1e-8 and abs((a @ b) / len(a) - lam[j]) < 1e-8


@pytest.mark.gpu
def test_g9b_device_weights_and_kl_reproduce_reference():
    from esmdiff_amd import metrics as M
    got = {"js_pwd_w": M.js_pwd(ENS, weights=dict(W), rounded=False), "js_rg_w": M.js_rg(ENS, weights=dict(W), rounded=False),
           "js_pwd_kl": M.js_pwd(ENS, kl=True, rounded=False), "js_rg_kl": M.js_rg(ENS, kl=True, rounded=False),
           "js_pwd_w_kl_b20": M.js_pwd(ENS, n_bins=20, weights=dict(W), kl=True, rounded=False)}
    for name, res in got.items():
        for i, k in enumerate(KEYS):
            assert abs(res[k] - GB[name + "_raw"][i]) < 1e-11, (name, k, res[k], GB[name + "_raw"][i])


@pytest.mark.gpu
def test_device_js_tica_follows_the_oracle_and_is_sign_scale_free():
    from esmdiff_amd import metrics as M
    rng = np.random.default_rng(5)
    ref = _trajectory(rng, 14, 600)
    mod = _trajectory(rng, 14, 300, lag_corr=0.9) * 1.02
    res, tic = M.js_tica({"target": ref, "m": mod}, lagtime=20, rounded=False)
    want = R.js_tica(mod, ref, 50, 20)
    assert res["target"] == 0.0 and abs(res["m"] - want) < 1e-6, (res, want)   # eigenvectors agree to ~1e-9 -> a few bin flips at most
    assert tic["target"].shape == (600, 2) and tic["m"].shape == (300, 2)
    w = {"m": rng.uniform(0.5, 1.5, 300)}
    rw = M.js_tica({"target": ref, "m": mod}, lagtime=20, return_tic=False, weights=w, rounded=False)["m"]
    assert abs(rw - R.js_tica(mod, ref, 50, 20, w_model=w["m"])) < 1e-6
    with pytest.raises(ValueError, match="lagtime"):
        M.js_tica({"target": ref[:15], "m": mod}, lagtime=20)
    assert M.js_tica({"target": ref, "m": mod}, lagtime=20, return_tic=False)["m"] == float(np.around(res["m"], 4))


# ---- g9c: the small helpers around the metrics (eval_utils.py:37-129, :191-224, :495-530) ---------------------------------
GC = np.load(Path(__file__).parent / "golden" / "g9c_ensemble_helpers.npz")
IDP = ("mse_pwd", "mse_rg", "mse_contact", "mae_pwd", "mae_rg", "mae_contact")


def test_g9c_oracle_reproduces_reference_helpers():
    import json
    for tag, k in (("", 3), ("_k1", 1)):
        got = R.idp_metrics(dict(ENS), pwd_offset=k)
        for nm, d in zip(IDP, got):
            want = GC[f"idp_{nm}{tag}"]
            assert np.allclose([d[key] for key in KEYS], want, rtol=1e-12, atol=1e-14), (nm, tag)
            assert d["target"] == 0.0
    assert np.allclose(R.rmsf(ENS["target"]), GC["rmsf_target"], rtol=1e-13, atol=0)
    assert np.array_equal(R.pairwise_adjacent(ENS["model_a"]), GC["adjacent_model_a"])
    assert np.allclose(R.distance_matrix_ca(ENS["model_b"]), GC["distance_matrix_model_b"], rtol=1e-15, atol=0)
    assert np.allclose(R.radius_of_gyration_masses(ENS["target"], GC["masses"]), GC["rg_target_masses"], rtol=1e-14, atol=0)
    pse = R.position_specific_entropy(GC["tokens"])
    assert np.allclose(pse, GC["position_specific_entropy"], rtol=0, atol=1e-6) and pse[2] == 0.0
    g = json.loads((Path(__file__).parent / "golden" / "g9c_split_pdb.json").read_text())
    merged = json.loads((Path(__file__).parent / "golden" / "g8_merge_pdb.json").read_text())["merged"]
    assert R.split_pdb_text(merged) == g["parts"] and len(g["parts"]) == 2


def test_g9c_split_pdbfile_inverts_merge(tmp_path):
    import json

    from esmdiff_amd.pdbio import merge_pdbfiles, split_pdbfile
    g = json.loads((Path(__file__).parent / "golden" / "g9c_split_pdb.json").read_text())
    g8 = json.loads((Path(__file__).parent / "golden" / "g8_merge_pdb.json").read_text())
    src = tmp_path / "merged.pdb"
    src.write_text(g8["merged"])
    parts = split_pdbfile(src, output_dir=tmp_path / "split", verbose=False)
    assert parts == g["parts"]
    assert {p.name: p.read_text() for p in sorted((tmp_path / "split").iterdir())} == g["files"]
    assert split_pdbfile(src, verbose=False) == g["parts"] and len(list((tmp_path / "split").iterdir())) == 2
    # round trip: merging the split files gives the merged file back
    merge_pdbfiles(sorted((tmp_path / "split").iterdir()), tmp_path / "again.pdb", verbose=False)
    assert (tmp_path / "again.pdb").read_text() == g8["merged"]
    with pytest.raises(AssertionError):
        split_pdbfile(tmp_path / "nope.pdb")


@pytest.mark.gpu
def test_g9c_device_helpers_reproduce_reference():
    from esmdiff_amd import metrics as M
    for tag, k in (("", 3), ("_k1", 1)):
        got = M.idp_metrics(dict(ENS), pwd_offset=k)
        for nm, d in zip(IDP, got):
            assert np.allclose([d[key] for key in KEYS], GC[f"idp_{nm}{tag}"], rtol=1e-11, atol=1e-13), (nm, tag)
    assert np.allclose(M.rmsf(ENS["target"]).cpu().numpy(), GC["rmsf_target"], rtol=1e-12, atol=0)
    assert np.allclose(M.adjacent_ca_distance(ENS["model_a"]).cpu().numpy(), GC["adjacent_model_a"], rtol=1e-14, atol=0)
    assert np.allclose(M.distance_matrix_ca(ENS["model_b"]).cpu().numpy(), GC["distance_matrix_model_b"], rtol=1e-14, atol=0)
    assert np.allclose(M.radius_of_gyration(ENS["target"], GC["masses"]).cpu().numpy(), GC["rg_target_masses"], rtol=1e-13, atol=0)
    assert np.allclose(M.radius_of_gyration(ENS["target"]).cpu().numpy(), G["rg_target"], rtol=1e-13, atol=0)
    pse = M.position_specific_entropy(torch.from_numpy(GC["tokens"])).cpu().numpy()
    assert pse.dtype == np.float32 and np.allclose(pse, GC["position_specific_entropy"], rtol=0, atol=1e-6) and pse[2] == 0.0
    with pytest.raises(AssertionError, match="masses"):
        M.radius_of_gyration(ENS["target"], GC["masses"][:5])
