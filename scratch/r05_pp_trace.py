"""Phase stamps of the 8-wave ping-pong attention kernel (-DED_ATTN_TRACE build): python scratch/build_variant.py attention pptrace -DED_ATTN_TRACE;
ESMDIFF_LIB=esmdiff_amd/lib/libesmdiff_hip_pptrace.so python scratch/r05_pp_trace.py"""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from esmdiff_amd import _native as N
from esmdiff_amd.config import ModelConfig
from esmdiff_amd.engine import Engine
from esmdiff_amd.weights import random_init_state_dict
B, L = int(os.environ.get("BB", 100)), int(os.environ.get("LL", 258))
cfg = ModelConfig(n_layers=1)
eng = Engine(cfg, random_init_state_dict(cfg, 3, device="cuda"), max_batch=B, max_len=L)
qkv = (torch.randn(B * L, 3 * cfg.d_model, device="cuda") * 1.5).to(torch.bfloat16)
w = torch.ones(cfg.d_model, device="cuda")
for _ in range(3):
    eng.attention(qkv, w, w, B, L)
torch.cuda.synchronize()
buf = np.zeros(512 * 4 * 32, dtype=np.uint64)
lib = N.lib()
lib.esmdiff_debug_attn_trace.argtypes = [ctypes.c_void_p]
assert lib.esmdiff_debug_attn_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(512, 8, 16).astype(np.int64)
nkt = (L + 63) // 64
n = 4 + 2 * nkt
r = t[:, :, :n]
ok = (r != 0).all(-1)
for half in (0, 1):
    x = r[:, 4 * half:4 * half + 4][ok[:, 4 * half:4 * half + 4]]
    d = np.diff(x, axis=1)
    names = ["issue DMA + Q", "wait landed + barrier"] + sum([[f"compute({j})", f"softmax({j})"] for j in range(nkt)], []) + ["last P.V"]
    print(f"half {half}: {len(x)} waves; lifetime {np.mean(x[:, -1] - x[:, 0]):.0f} ticks (100 MHz s_memtime ticks if < 1e4, else shader cycles)")
    print("   " + "; ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d.mean(0))))
eng.close()
