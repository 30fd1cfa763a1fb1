"""30 seeds of certified sampling at configs[1] against the F32_SPLIT engine's own chain (FAST_RERUNS=1: K-sliced re-runs)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from esmdiff_amd.certified import CertifiedSampler
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict
sd = random_init_state_dict(cfg, seed=11, device="cuda")
B, L, T = 100, 258, 25
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
exact = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
fast = Engine(cfg, sd, max_batch=B, max_len=L, precision="f16", head_precision="f32")
cs = CertifiedSampler(fast, exact, fast_reruns=os.environ.get("FAST_RERUNS") == "1")
bad = 0
t0 = time.time()
for k in range(30):
    got = cs.ddpm_sample(seq, sch, seed=100 + k)
    want = exact.ddpm_sample(seq, sch, seed=100 + k)
    ok = bool(torch.equal(got, want))
    bad += not ok
    print(k, ok, cs.stats["sample_forwards_exact"], cs.stats["eps_violations"], round(cs.stats["max_logit_err_observed"], 5), flush=True)
print("certified soak: 30 seeds, mismatching runs:", bad, "elapsed", round(time.time() - t0, 1), "s")
