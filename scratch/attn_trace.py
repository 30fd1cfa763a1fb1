"""Where a wave of the attention kernel spends its time: s_memtime stamps of the -DED_ATTN_TRACE build.
    python scratch/build_variant.py attention trace -DED_ATTN_TRACE      (CPU container)
    ESMDIFF_LIB=$PWD/esmdiff_amd/lib/libesmdiff_hip_trace.so python scratch/attn_trace.py    (GPU box)
Stamps per wave: [kernel entry] then per key tile [tile start, scores + max ready, softmax + P.V done, own DMA of the next tile
landed, past the barrier], then [before the epilogue]."""
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
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); eng.attention(qkv, w, w, B, L); ev1.record(); torch.cuda.synchronize()
buf = np.zeros(512 * 4 * 32, dtype=np.uint64)
lib = N.lib()
lib.esmdiff_debug_attn_trace.argtypes = [ctypes.c_void_p]
assert lib.esmdiff_debug_attn_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(512, 4, 32).astype(np.int64)
nkt = (L + 63) // 64
n_full = 2 + 5 * nkt - 0            # entry + 5 per tile + pre-epilogue  (the last tile has no 'landed' wait difference: still stamped)
rows = []
for wg in range(512):
    for wv in range(4):
        s = t[wg, wv]
        k = int((s != 0).sum())
        if k < 2 + 5 * nkt:      # parked waves (no score stamp) and phantom blocks
            continue
        rows.append(s[: 2 + 5 * nkt])
r = np.array(rows)
print(f"B={B} L={L}: rope+attention {ev0.elapsed_time(ev1) * 1e3:.1f} us; {len(r)} working waves traced; ticks = s_memtime = shader cycles (a 27.8 k-tick wave lifetime is ~12.6 us)" )
life = r[:, -1] - r[:, 0]
print(f"  wave lifetime entry -> epilogue: mean {life.mean():.0f} ticks, p10 {np.percentile(life,10):.0f}, p90 {np.percentile(life,90):.0f}")
print(f"  prologue (entry -> first tile start): mean {(r[:,1]-r[:,0]).mean():.0f}")
names = ["QK^T + max (start -> scores ready)", "softmax + P.V", "wait own DMA of next tile", "barrier"]
for kt in range(nkt):
    b = 1 + 5 * kt
    seg = [r[:, b + 1] - r[:, b], r[:, b + 2] - r[:, b + 1], r[:, b + 3] - r[:, b + 2], r[:, b + 4] - r[:, b + 3]]
    print(f"  tile {kt}: " + "; ".join(f"{n} {x.mean():.0f}" for n, x in zip(names, seg)) + f"; total {(r[:, b + 4] - r[:, b]).mean():.0f}")
tot = np.zeros(4)
for kt in range(nkt):
    b = 1 + 5 * kt
    for j in range(4):
        tot[j] += (r[:, b + j + 1] - r[:, b + j]).mean()
print("  sum over tiles: " + "; ".join(f"{n} {x:.0f} ({100 * x / life.mean():.0f} %)" for n, x in zip(names, tot)))
eng.close()
