"""Cost of one re-run of certified sampling at configs[1] (run on the GPU box): the F32_SPLIT forward at B = 1 .. 16 samples of
258 tokens, and the host-side pieces around it (gather, error monitor, per-sample draws)."""
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
L, T = 258, 25
g = torch.Generator().manual_seed(258)
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
exact = Engine(cfg, sd, max_batch=100, max_len=L, precision="f32_split")
if os.environ.get("SPLITK") == "1":          # K-sliced residual linears at <= 4096 rows (esmdiff_set_small_batch_splitk)
    exact.set_small_batch_splitk(True)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in (1, 2, 4, 8, 10, 12, 16, 24, 32, 100):
    seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
    x = torch.full((B, L), 4096, dtype=torch.int64, device="cuda")
    x[:, ::2] = 7
    ms = timed(lambda: exact.forward_logits(x, seq, sch.t_freq[3]))
    lg = exact.forward_logits(x, seq, sch.t_freq[3])
    lg_b = lg.clone()
    mon = timed(lambda: float(((lg - lg_b).abs().amax(-1) * (x == 4096)).amax(-1).max()))
    step = timed(lambda: [exact.ddpm_step(x[j:j + 1].clone(), lg[j:j + 1], 0.5, 0.4, seed=1, sample_offset=j, step=2) for j in range(B)])
    print(f"B={B:3d} ({B * L:5d} rows): forward {ms:7.2f} ms = {ms / B:5.2f} ms per sample; monitor {mon:5.2f} ms; {B} single-sample draws {step:5.2f} ms", flush=True)
