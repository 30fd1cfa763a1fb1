"""Prototype: does running two half-batches on two HIP streams fill the tile-quantisation tails?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.weights import random_init_state_dict
from esmdiff_amd.schedule import timestep_embedding
dev = torch.device("cuda:0")
sd = random_init_state_dict(cfg, seed=0, device="cuda:0")
L = 258
def mk(B):
    e = Engine(cfg, sd, max_batch=B, max_len=L, device=0)
    seq = torch.randint(4, 24, (B, L), device=dev); seq[:, 0] = 0; seq[:, -1] = 2
    x = torch.full((B, L), 4096, device=dev, dtype=torch.int64)
    tf = timestep_embedding(torch.tensor([3.0]), cfg.freq_dim).to(dev)
    return e, seq, x, tf
full = mk(100)
h1, h2 = mk(50), mk(50)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_full(n):
    for _ in range(n): full[0].forward_logits(full[2], full[1], full[3])
def run_half(n):
    for _ in range(n):
        with torch.cuda.stream(s1): h1[0].forward_logits(h1[2], h1[1], h1[3])
        with torch.cuda.stream(s2): h2[0].forward_logits(h2[2], h2[1], h2[3])
for name, fn in (("one stream B=100", run_full), ("two streams B=50+50", run_half), ("one stream B=100", run_full), ("two streams B=50+50", run_half)):
    fn(2); torch.cuda.synchronize(); t = time.perf_counter(); fn(8); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 8 * 1e3:.2f} ms per 100-sample forward")
