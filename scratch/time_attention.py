"""Per-launch time of qk_norm_rope + attention at BASELINE configs[1] / [3] shapes (HIP events around 50 calls)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd.config import ModelConfig
from esmdiff_amd.engine import Engine
from esmdiff_amd.weights import random_init_state_dict
cfg = ModelConfig(n_layers=1)
sd = random_init_state_dict(cfg, seed=0, device="cuda:0")
for B, L in ((100, 258), (32, 1026)):
    eng = Engine(cfg, sd, max_batch=B, max_len=L)
    qkv = torch.randn(B * L, 3 * cfg.d_model, device="cuda").to(torch.bfloat16)
    w = torch.ones(cfg.d_model, device="cuda")
    for _ in range(5):
        eng.attention(qkv, w, w, B, L)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        eng.attention(qkv, w, w, B, L)
    b.record()
    torch.cuda.synchronize()
    print(f"B={B} L={L}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per (qk_norm_rope + attention)")
    eng.close()
