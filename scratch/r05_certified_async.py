"""r05: the speculative / batched / audited certified sampler at configs[1] (100 x 258, 25 updates, 48 blocks).

  python scratch/r05_certified_async.py [--seeds N] [--sweep]

Prints (and writes to gpurun_out/r05_certified_async.json): forward time of the F32_SPLIT engine vs batch size with one sigma per
sample (what a verification batch costs), the fast engine's small-batch forward (what a roll-back tail costs), then the certified
sampler cold + warm with its stats, ids against the F32_SPLIT engine's own chain."""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd.certified import CertifiedSampler      # noqa: E402
from esmdiff_amd.config import ESM3_OPEN                # noqa: E402
from esmdiff_amd.engine import Engine                   # noqa: E402
from esmdiff_amd.schedule import ddpm_schedule          # noqa: E402
from esmdiff_amd.weights import random_init_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=3)
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--B", type=int, default=100)
ap.add_argument("--L", type=int, default=258)
ap.add_argument("--steps", type=int, default=25)
args = ap.parse_args()

cfg = ESM3_OPEN
B, L, T = args.B, args.L, args.steps
sd = random_init_state_dict(cfg, seed=11, device="cuda")
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
exact = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
fast = Engine(cfg, sd, max_batch=B, max_len=L, precision="f16", head_precision="f32")
out = {"B": B, "L_tok": L, "steps": T}


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


x0 = torch.full((B, L), 4096, dtype=torch.int64, device="cuda")
x0[:, ::3] = 17
tfe = exact.conditioning_rows(sch.t_freq).cuda()
tff = fast.conditioning_rows(sch.t_freq).cuda()
rows = torch.arange(B, device="cuda") % (T + 1)
out["f32_split_forward_ms"] = {}
for n in (4, 8, 16, 24, 32, 48, 64, B):
    if n <= B:
        out["f32_split_forward_ms"][n] = round(timed(lambda: exact.forward_logits(x0[:n], seq[:n], tfe[rows[:n]])), 2)
out["fast_forward_ms"] = {}
for n in (1, 2, 4, 8, 16, B):
    if n <= B:
        out["fast_forward_ms"][n] = round(timed(lambda: fast.forward_logits(x0[:n], seq[:n], tff[rows[:n]] if n > 1 else tff[0]), 5), 2)
print(json.dumps(out), flush=True)

runs = []
variants = [dict()]
if args.sweep:
    variants += [dict(verify_batch=16), dict(verify_batch=48), dict(k_sigma=5.0), dict(k_sigma=8.0), dict(audit_rate=0.0)]
for kw in variants:
    cs = CertifiedSampler(fast, exact, **kw)
    for s_ in range(args.seeds):
        seed = 23 + s_
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got = cs.ddpm_sample(seq, sch, seed=seed)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        want = exact.ddpm_sample(seq, sch, seed=seed)
        st = dict(cs.stats)
        rec = {"kw": kw, "seed": seed, "seconds": round(dt, 3), "samples_per_s": round(B / dt, 2),
               "ids_equal_to_f32_split_chain": bool(torch.equal(got, want)),
               "samples_identical": int((got == want).all(1).sum()), **st}
        runs.append(rec)
        print(json.dumps(rec), flush=True)
out["runs"] = runs
plain = fast.ddpm_sample(seq, sch, seed=23)
out["uncertified_fast_samples_identical_seed23"] = int((plain == exact.ddpm_sample(seq, sch, seed=23)).all(1).sum())
p = Path(__file__).resolve().parent.parent / "gpurun_out" / "r05_certified_async.json"
p.parent.mkdir(exist_ok=True)
p.write_text(json.dumps(out, indent=1))
print("written", p)
