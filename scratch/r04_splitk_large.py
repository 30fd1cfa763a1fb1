"""F32_SPLIT at configs[1] (100 x 258 tokens) with the residual linears K-sliced at EVERY size (ESMDIFF_SPLITK_MAX_ROWS=1000000)
against the default: forward time, chain time, ids.  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict

sd = random_init_state_dict(cfg, seed=11, device="cuda")
B, L, T = 100, 258, 25
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
eng = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
x = torch.full((B, L), 4096, dtype=torch.int64, device="cuda")


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for on in (False, True, False, True):
    eng.set_small_batch_splitk(on)
    tf = timed(lambda: eng.forward_logits(x, seq, sch.t_freq[3]), 5)
    tc = timed(lambda: eng.ddpm_sample(seq, sch, seed=1), 2)
    ids = eng.ddpm_sample(seq, sch, seed=1)
    if not on:
        ref = ids
    print(f"K-sliced {on}: forward {tf * 1e3:.1f} ms, chain {tc:.3f} s = {B / tc:.2f} samples/s, ids equal to the unsliced chain: {bool(torch.equal(ids, ref))}", flush=True)
