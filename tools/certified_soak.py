"""Measured bounds for certified sampling: N jobs of configs[1]'s size (100 samples x 258 tokens) against the
F32_SPLIT engine's own chain, one line per job, and at the end what the MEASUREMENT supports — no model of the error:

  * jobs identical / jobs run, and the rule-of-three upper bound (95 %) on the job-level miss rate when none missed: 3 / jobs
  * audits clean / audits run, and the same bound on the miss rate of an unflagged sample-update: 3 / audits
  * violations of the bounds in use (a verified row whose error RANGE exceeded P, or whose entropy error exceeded E)
  * samples/s of the certified sampler and of the F32_SPLIT engine alone

    python tools/certified_soak.py --mode ddpm|gibbs [--weights random|trained_like] [--jobs 200] [--steps 25]
                                   [--inpaint]   (gibbs: configs[4]'s shape — 64 masked residues, backbone frames for the rest)
                                   [--out gpurun_out/soak.txt]

The output is appended to --out as it is produced (a job takes ~8 s: a killed run keeps what it measured).  Runs on the GPU
box; imports nothing from oracle/ (the referee is the F32_SPLIT engine, whose agreement with the float32 oracle chain is the
subject of tests/test_gpu_strict.py)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esmdiff_amd.certified import CertifiedSampler       # noqa: E402
from esmdiff_amd.config import ESM3_OPEN as cfg          # noqa: E402
from esmdiff_amd.engine import Engine                    # noqa: E402
from esmdiff_amd.schedule import ddpm_schedule           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["ddpm", "gibbs"], default="ddpm")
ap.add_argument("--weights", choices=["random", "trained_like"], default="random")
ap.add_argument("--jobs", type=int, default=30)
ap.add_argument("--steps", type=int, default=25)
ap.add_argument("--first_seed", type=int, default=1000)
ap.add_argument("--inpaint", action="store_true")
ap.add_argument("--out", default="gpurun_out/certified_soak.txt")
ap.add_argument("--budget_s", type=float, default=1e9, help="stop starting new jobs after this many seconds")
ap.add_argument("--direct_share", type=float, default=None, help="diagnostics: CertifiedSampler(direct_share=...) (1.0: never the direct lane)")
ap.add_argument("--audit_rate", type=float, default=None, help="diagnostics: audit this share of the unflagged sample-updates throughout")
a = ap.parse_args()

os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
fh = open(a.out, "a")


def say(*parts):
    line = " ".join(str(p) for p in parts)
    print(line, flush=True)
    fh.write(line + "\n")
    fh.flush()


t_start = time.perf_counter()
if a.weights == "random":
    from esmdiff_amd.weights import random_init_state_dict
    sd = random_init_state_dict(cfg, seed=11, device="cuda", with_geom=True)
else:
    from esmdiff_amd.weights import trained_like_state_dict
    sd = trained_like_state_dict(cfg, seed=11, device="cuda", with_geom=True)
B, L, T = 100, 258, a.steps
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
exact = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
fast = Engine(cfg, sd, max_batch=B, max_len=L, precision="f16", head_precision="f32")
del sd
kw = {}
if a.direct_share is not None:
    kw["direct_share"] = a.direct_share
if a.audit_rate is not None:
    kw.update(audit_rate=a.audit_rate, audit_rate_steady=a.audit_rate)
cs = CertifiedSampler(fast, exact, **kw)
frames = None
if a.mode == "ddpm":
    sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
    run_cert = lambda seed: cs.ddpm_sample(seq, sch, seed=seed)
    run_exact = lambda seed: exact.ddpm_sample(seq, sch, seed=seed)
else:
    from esmdiff_amd.gibbs import unmask_schedule
    x0 = torch.full((B, L), 4096, dtype=torch.int64)
    x0[:, 0], x0[:, -1] = 4098, 4097
    n_masked = L - 2
    if a.inpaint:          # configs[4]: residues 96..159 are sampled, the rest of the backbone conditions block 0's geometric attention
        from esmdiff_amd.geometry import build_affine3d_from_coordinates
        n_masked = 64
        x0[:, 1:-1] = torch.randint(0, 4096, (1, L - 2), generator=g)       # known structure tokens outside the window
        x0[:, 97:161] = 4096
        ca = torch.cumsum(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1) * 3.8, 0)
        xyz = torch.stack([ca + torch.tensor([-1.2, 0.7, 0.0]), ca, ca + torch.tensor([1.3, 0.6, 0.1])], 1)
        xyz[97:161] = float("inf")
        xyz[0] = xyz[-1] = float("nan")
        frames = build_affine3d_from_coordinates(xyz[None].repeat(B, 1, 1, 1))
    table = torch.tensor(unmask_schedule(n_masked, T), dtype=torch.int32)[:, None].repeat(1, B)

    def run_cert(seed):
        return cs.gibbs_sample(seq, x0, table, 1.4, 0.9, seed=seed, frames=frames)

    def run_exact(seed):
        if frames is not None:
            exact.set_frames(*frames)
        try:
            return exact.gibbs_sample(seq, x0, table, 1.4, 0.9, seed=seed)
        finally:
            if frames is not None:
                exact.set_frames(None)

say(f"# certified soak: mode={a.mode} weights={a.weights} inpaint={a.inpaint} B={B} L_tok={L} steps={T} jobs<={a.jobs} first_seed={a.first_seed} "
    f"certificate='{cs.stats.get('certificate', 'k-sigma statistical + audit')}' k_sigma={cs.k_sigma} audit {cs.audit_rate} -> {cs.audit_rate_steady} after {cs.audit_clean_target} clean")
say("seed ids_equal samples_differing cert_s exact_s flagged race nucleus order corrections audit_checked audit_mismatches eps_viol entropy_viol P_used E_used max_range_err max_entropy_err audit_rate fwd_fast fwd_direct lane_switches")
tot = {k: 0 for k in ("jobs", "jobs_equal", "samples_differing", "flagged", "corrections", "audit_checked", "audit_mismatches", "eps_violations",
                      "entropy_violations", "sample_updates", "sample_forwards_fast", "sample_forwards_exact")}
t_cert = t_exact = 0.0
for k in range(a.jobs):
    if time.perf_counter() - t_start > a.budget_s:
        say(f"# time budget reached after {k} jobs")
        break
    seed = a.first_seed + k
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = run_cert(seed)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = cs.stats
    torch.cuda.synchronize(); t0 = time.perf_counter()
    want = run_exact(seed)
    torch.cuda.synchronize(); de = time.perf_counter() - t0
    ok = bool(torch.equal(got, want))
    ndiff = int((got != want).any(1).sum())
    if not ok:
        bad = torch.nonzero((got != want).any(1)).flatten().tolist()
        say(f"#   seed {seed}: samples {bad} differ at positions {[torch.nonzero(got[b] != want[b]).flatten().tolist()[:8] for b in bad]}")
    if k:
        t_cert += dt
        t_exact += de
    tot["jobs"] += 1
    tot["jobs_equal"] += int(ok)
    tot["samples_differing"] += ndiff
    for key in ("flagged", "corrections", "audit_checked", "audit_mismatches", "eps_violations", "entropy_violations", "sample_forwards_fast",
                "sample_forwards_exact"):
        tot[key] += st[key]
    fr = st["flag_reasons"]
    say(seed, ok, ndiff, round(dt, 3), round(de, 3), st["flagged"], fr["race"], fr["nucleus"], fr["order"], st["corrections"], st["audit_checked"],
        st["audit_mismatches"], st["eps_violations"], st["entropy_violations"], f"{2 * (st['eps_max_used'] or 0):.3e}", f"{st.get('entropy_eps_max_used', 0) or 0:.3e}",
        f"{st['max_range_err_observed']:.3e}", f"{st['max_entropy_err_observed']:.3e}", st["audit_rate_now"], st["sample_forwards_fast"],
        st["sample_forwards_direct"], st["direct_lane_switches"])
n = max(tot["jobs"], 1)
r3 = lambda m: f"{3.0 / m:.2e}" if m else "n/a"
say(f"# totals: {tot}")
say(f"# jobs identical to the F32_SPLIT chain: {tot['jobs_equal']} / {tot['jobs']}"
    + (f"; rule of three: job-level miss rate < {r3(tot['jobs'])} (95 %)" if tot["jobs_equal"] == tot["jobs"] else "; MISSES OBSERVED"))
say(f"# audits clean: {tot['audit_checked'] - tot['audit_mismatches']} / {tot['audit_checked']}"
    + (f"; rule of three: miss rate of an unflagged sample-update < {r3(tot['audit_checked'])} (95 %)" if tot["audit_mismatches"] == 0 else "; AUDIT MISSES OBSERVED"))
say(f"# bound violations: pair {tot['eps_violations']}, entropy {tot['entropy_violations']}; flagged share {tot['flagged'] / max(tot['sample_forwards_fast'], 1):.4f} "
    f"of the fast sample-forwards; verified {tot['sample_forwards_exact']}")
say(f"# measured error of this engine pair: sigma_pair {cs.sigma_d_seen:.3e}, largest row range {cs.range_seen:.3e}, largest |e| {cs.err_seen:.3e}, "
    f"sigma_entropy {cs.sigma_h_seen:.3e}, largest entropy error {cs.max_dh_seen:.3e}, items {cs.n_seen}")
if tot["jobs"] > 1:
    say(f"# samples/s over jobs 2..{tot['jobs']} (the first is cold): certified {B * (tot['jobs'] - 1) / max(t_cert, 1e-9):.2f}, "
        f"F32_SPLIT alone {B * (tot['jobs'] - 1) / max(t_exact, 1e-9):.2f}, ratio {t_exact / max(t_cert, 1e-9):.2f}")
fh.close()
