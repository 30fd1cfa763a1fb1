"""r05: N seeds (default 30) of certified sampling at configs[1] against the F32_SPLIT engine's own chain; one line per seed +
the totals VERDICT r04 item 1 asks for (ids equal, audit_checked / audit_mismatches / audit_max_logit_err, samples/s)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from esmdiff_amd.certified import CertifiedSampler
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sd = random_init_state_dict(cfg, seed=11, device="cuda")
B, L, T = 100, 258, 25
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
exact = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
fast = Engine(cfg, sd, max_batch=B, max_len=L, precision="f16", head_precision="f32")
cs = CertifiedSampler(fast, exact)
bad, tot, t_cert = 0, {}, 0.0
print("seed ids_equal seconds gpu_fast gpu_verify tail fast_launches flagged corrections audit_checked audit_mismatches eps_violations eps_used max_pair_err")
for k in range(n_seeds):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = cs.ddpm_sample(seq, sch, seed=100 + k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    want = exact.ddpm_sample(seq, sch, seed=100 + k)
    ok = bool(torch.equal(got, want))
    bad += not ok
    st = cs.stats
    if k:
        t_cert += dt
    for key in ("flagged", "corrections", "audit_checked", "audit_mismatches", "audit_eps_violations", "eps_violations",
                "sample_forwards_fast", "sample_forwards_exact", "rollback_updates_discarded"):
        tot[key] = tot.get(key, 0) + st[key]
    tot["audit_max_logit_err"] = max(tot.get("audit_max_logit_err", 0.0), st["audit_max_logit_err"])
    tot["audit_max_pair_err"] = max(tot.get("audit_max_pair_err", 0.0), st["audit_max_pair_err"])
    print(100 + k, ok, round(dt, 3), st.get("gpu_seconds_fast"), st.get("gpu_seconds_verify"), st["tail_seconds"], st["fast_launches"], st["flagged"], st["corrections"], st["audit_checked"], st["audit_mismatches"], st["eps_violations"],
          f"{st['eps_max_used']:.3e}", f"{st['max_pair_err_observed']:.3e}", flush=True)
print(f"certified soak: {n_seeds} seeds x {B} samples, mismatching runs: {bad}; totals {tot}")
N5 = 5 * B
torch.cuda.synchronize(); t0 = time.perf_counter()
got = cs.ddpm_sample(seq[:1].expand(N5, L).contiguous(), sch, seed=999)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ok5 = all(bool(torch.equal(got[k * B:(k + 1) * B], exact.ddpm_sample(seq, sch, seed=999, sample_offset=k * B))) for k in range(5))
print(f"streamed: {N5} samples in one call, lane width {B}: {N5 / dt:.2f} samples/s, ids equal to the F32_SPLIT chain: {ok5}, "
      f"audit {cs.stats['audit_checked']} checked / {cs.stats['audit_mismatches']} mismatches, tail {cs.stats['tail_seconds']} s")
print(f"sigma_pair_err {cs.sigma_d_seen:.3e} max_pair_err {cs.max_d_seen:.3e} max_logit_err {cs.err_seen:.3e} items_seen {cs.n_seen}; "
      f"samples/s over seeds 2..{n_seeds} (the first call is cold): {B * (n_seeds - 1) / max(t_cert, 1e-9):.2f}")
