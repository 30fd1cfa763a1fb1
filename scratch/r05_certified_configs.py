"""Certified sampling at the other BASELINE configs (run on the GPU box): ids against the F32_SPLIT engine's own chain, re-run
share and rate.  configs[3]: 32 samples x 1024 residues, 25 updates; configs[4]: 100 x 256, 50 updates, residues 96..159 masked
(inpainting prior); configs[0]: 4 x 58.  -> gpurun_out/r05_certified_configs.json (r05 sampler: speculative lane, batched verification, audit)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from esmdiff_amd.certified import CertifiedSampler
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict

sd = random_init_state_dict(cfg, seed=11, device="cuda")
out = {}
for name, B, R, T, window in (("configs0", 4, 58, 25, None), ("configs3", 32, 1024, 25, None), ("configs4", 100, 256, 50, (96, 160))):
    L = R + 2
    g = torch.Generator().manual_seed(R)
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (R,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
    sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
    prior = None
    if window:
        prior = torch.randint(0, 4096, (1, L), generator=g).repeat(B, 1)
        prior[:, 0], prior[:, -1] = 4098, 4097
        prior[:, window[0] + 1:window[1] + 1] = 4096
        prior = prior.cuda()
    exact = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
    fast = Engine(cfg, sd, max_batch=B, max_len=L, precision="f16", head_precision="f32")
    cs = CertifiedSampler(fast, exact)
    cold = cs.ddpm_sample(seq, sch, seed=5, input_prior=prior)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = cs.ddpm_sample(seq, sch, seed=5, input_prior=prior)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    want = exact.ddpm_sample(seq, sch, seed=5, input_prior=prior)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    plain = fast.ddpm_sample(seq, sch, seed=5, input_prior=prior)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    st = cs.stats
    out[name] = {"B": B, "L_tok": L, "updates": T, "masked_window": window,
                 "ids_equal_to_f32_split_chain": bool(torch.equal(got, want)), "cold_call_equal": bool(torch.equal(cold, want)),
                 "samples_identical_without_certification": int((plain == want).all(1).sum()),
                 "certified_samples_per_s": round(B / (t1 - t0), 2), "f32_split_samples_per_s": round(B / (t2 - t1), 2),
                 "f16_head_f32_samples_per_s": round(B / (t3 - t2), 2),
                 "rerun_share": st["rerun_share"], "exact_per_fast_sample_forwards": round(st["sample_forwards_exact"] / max(1, st["sample_forwards_fast"]), 4),
                 "eps_used": [st["eps_min_used"], st["eps_max_used"]], "max_logit_err_observed": st["max_logit_err_observed"],
                 "eps_violations": st["eps_violations"], "first_update_shared": st["first_update_shared"],
                 "flagged": st["flagged"], "corrections": st["corrections"], "audit_checked": st["audit_checked"],
                 "audit_mismatches": st["audit_mismatches"], "verify_batch_sizes": st["verify_batch_sizes"], "tail_seconds": st["tail_seconds"],
                 "gpu_seconds_fast": st.get("gpu_seconds_fast"), "gpu_seconds_verify": st.get("gpu_seconds_verify")}
    print(name, json.dumps(out[name]), flush=True)
    fast.close(); exact.close()
json.dump(out, open("gpurun_out/r05_certified_configs.json", "w"), indent=1)
