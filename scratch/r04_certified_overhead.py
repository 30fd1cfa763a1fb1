"""Where certified sampling's time goes at configs[1] (run on the GPU box): the fast engine's device loop, the same chain driven
step by step by certified.py with (almost) nothing flagged (eps = 1e-9), and the certified run."""
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
B, L, T = 100, 258, 25
g = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
exact = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
fast = Engine(cfg, sd, max_batch=B, max_len=L, precision="f16", head_precision="f32")


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


fast.set_step0_sharing(True); fast.set_final_skip(True)
t_dev = timed(lambda: fast.ddpm_sample(seq, sch, seed=1))
fast.set_step0_sharing(False); fast.set_final_skip(False)
none = CertifiedSampler(fast, exact, eps=1e-9)
t_none = timed(lambda: none.ddpm_sample(seq, sch, seed=1))
cs = CertifiedSampler(fast, exact)
t_cert = timed(lambda: cs.ddpm_sample(seq, sch, seed=1))
print(f"fast device loop with the two exact shortcuts {t_dev:.3f} s; step-wise drive, nothing flagged {t_none:.3f} s "
      f"({none.stats['sample_forwards_exact']} exact sample-forwards); certified {t_cert:.3f} s "
      f"({cs.stats['sample_forwards_exact']} exact sample-forwards)")
