"""r05: every precision on TRAINED-LIKE synthetic weights (esmdiff_amd.weights.trained_like_state_dict) at the full 48 blocks:
logit scale, finite-ness, logit error of each engine against the exact-f32 engine, id chains, certified sampling from a cold start.
Writes gpurun_out/r05_trained_like.json."""
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from esmdiff_amd.certified import CertifiedSampler
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import trained_like_state_dict
B, L, T = int(os.environ.get("B", 100)), 258, 25
sd = trained_like_state_dict(cfg, seed=11, device="cuda")
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
out = {"B": B, "L_tok": L, "steps": T, "weights": "trained_like_state_dict(ESM3_OPEN, seed=11)"}
eng = {}
for name, kw in (("f32", {"precision": "f32"}), ("f32_split", {"precision": "f32_split"}), ("f16_f32head", {"precision": "f16", "head_precision": "f32"}),
                 ("f16", {"precision": "f16"}), ("bf16", {})):
    eng[name] = Engine(cfg, sd, max_batch=B, max_len=L, **kw)
n = 8
x = torch.full((n, L), 4096, dtype=torch.int64)
x[:, 1::3] = torch.randint(0, 4096, (n, len(range(1, L, 3))), generator=g)
x = x.cuda()
lg = {k: e.forward_logits(x, seq[:n], sch.t_freq[5]).float().clone() for k, e in eng.items()}
ref = lg["f32"]
out["logits"] = {"std": float(ref.std()), "absmax": float(ref.abs().max()), "mean_max_prob": float(torch.softmax(ref, -1).max(-1).values.mean())}
for k, v in lg.items():
    d = (v - ref)
    out["logits"][k] = {"finite": bool(torch.isfinite(v).all()), "max_err_vs_f32": float(d.abs().max()), "rms_err_vs_f32": float(d.pow(2).mean().sqrt())}
print(json.dumps(out["logits"]), flush=True)
emb = eng["f32"].embeddings(n, L)
out["residual_absmax"] = float(emb.abs().max())
# chains
t0 = time.perf_counter(); c32 = eng["f32"].ddpm_sample(seq, sch, seed=23); torch.cuda.synchronize(); out["f32_seconds"] = round(time.perf_counter() - t0, 2)
t0 = time.perf_counter(); csp = eng["f32_split"].ddpm_sample(seq, sch, seed=23); torch.cuda.synchronize(); out["f32_split_seconds"] = round(time.perf_counter() - t0, 2)
out["f32_split_chain_equals_f32_chain"] = bool(torch.equal(c32, csp))
out["f32_split_samples_identical"] = int((c32 == csp).all(1).sum())
masked_draws = B * L
for k in ("f16_f32head", "f16", "bf16"):
    c = eng[k].ddpm_sample(seq, sch, seed=23)
    out[k + "_samples_identical_to_f32_split_chain"] = int((c == csp).all(1).sum())
    out[k + "_ids_differing"] = int((c != csp).sum())
print(json.dumps({k: v for k, v in out.items() if k != "logits"}), flush=True)
for fast_name in ("f16_f32head", "f16"):
    cs = CertifiedSampler(eng[fast_name], eng["f32_split"])
    recs = []
    for call in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got = cs.ddpm_sample(seq, sch, seed=23 + call)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        want = csp if call == 0 else eng["f32_split"].ddpm_sample(seq, sch, seed=23 + call)
        st = cs.stats
        recs.append({"call": call, "seconds": round(dt, 3), "samples_per_s": round(B / dt, 2), "ids_equal_to_f32_split_chain": bool(torch.equal(got, want)),
                     **{k: st[k] for k in ("flagged", "corrections", "audit_checked", "audit_mismatches", "audit_eps_violations", "eps_violations",
                                           "eps_min_used", "eps_max_used", "sigma_pair_err", "max_pair_err_observed", "max_logit_err_observed",
                                           "rerun_share", "sample_forwards_fast", "sample_forwards_exact", "fast_launches", "tail_seconds")}})
        print(fast_name, json.dumps(recs[-1]), flush=True)
    out["certified_" + fast_name] = recs
open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "r05_trained_like.json"), "w").write(json.dumps(out, indent=1))
