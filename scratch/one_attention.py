import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd.config import ModelConfig
from esmdiff_amd.engine import Engine
from esmdiff_amd.weights import random_init_state_dict
cfg = ModelConfig(n_layers=1)
B, L = int(os.environ.get("BB", 100)), int(os.environ.get("LL", 258))
sd = random_init_state_dict(cfg, seed=0, device="cuda:0")
eng = Engine(cfg, sd, max_batch=B, max_len=L)
qkv = torch.randn(B * L, 3 * cfg.d_model, device="cuda").to(torch.bfloat16)
w = torch.ones(cfg.d_model, device="cuda")
for _ in range(3):
    eng.attention(qkv, w, w, B, L)
torch.cuda.synchronize()
print("done")
