"""Does hipGraph replay help the small-batch forward?  (torch.cuda.graph captures the engine's launches.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.weights import random_init_state_dict
from esmdiff_amd.schedule import timestep_embedding
dev = torch.device("cuda:0")
sd = random_init_state_dict(cfg, seed=0, device="cuda:0")
for B, L in ((4, 60), (1, 258), (16, 258)):
    e = Engine(cfg, sd, max_batch=B, max_len=L, device=0)
    seq = torch.randint(4, 24, (B, L), device=dev); seq[:, 0] = 0; seq[:, -1] = 2
    x = torch.full((B, L), 4096, device=dev, dtype=torch.int64)
    tf = timestep_embedding(torch.tensor([3.0]), cfg.freq_dim).to(dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): e.forward_logits(x, seq, tf)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): e.forward_logits(x, seq, tf)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t) / 20
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            e.forward_logits(x, seq, tf)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t) / 20
    print(f"B={B} L={L}: eager {eager*1e3:.2f} ms/forward, graph replay {graph*1e3:.2f} ms/forward")
    e.close()
