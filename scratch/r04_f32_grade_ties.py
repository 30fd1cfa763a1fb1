"""How often two float32-grade evaluations of the same chain part ways: the exact-f32 engine (v_mfma_f32_32x32x2_f32, fmaf order)
against the F32_SPLIT engine (three f16 MFMA passes), configs[1] (100 x 258 tokens, 25 updates), 16 seeds.  Their logits differ
by up to ~7e-6; a draw tied to within that can go either way in either engine — as it can between the reference's own runs at
two batch splits."""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict

sd = random_init_state_dict(cfg, seed=11, device="cuda")
B, L, T = 100, 258, 25
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
strict = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32")
split = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
runs = samples = ids = 0
N = int(os.environ.get("SEEDS", "16"))
t0 = time.time()
for seed in range(200, 200 + N):
    a = strict.ddpm_sample(seq, sch, seed=seed)
    b = split.ddpm_sample(seq, sch, seed=seed)
    d = a != b
    runs += bool(d.any()); samples += int(d.any(1).sum()); ids += int(d.sum())
    print(seed, "equal" if not bool(d.any()) else f"{int(d.any(1).sum())} samples / {int(d.sum())} ids differ", flush=True)
print(f"exact-f32 vs F32_SPLIT chains over {N} seeds: {runs} runs differ ({samples} of {N * B} samples, {ids} ids); {time.time() - t0:.0f} s")
