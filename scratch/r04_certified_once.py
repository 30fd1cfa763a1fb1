"""One certified sampling run at configs[1] (100 x 258 tokens, 25 updates) for a kernel trace:
rocprofv3 --kernel-trace --stats -- python scratch/r04_certified_once.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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
cs = CertifiedSampler(fast, exact)
for seed in (1, 2):
    x = cs.ddpm_sample(seq, sch, seed=seed)
torch.cuda.synchronize()
print(cs.stats)
