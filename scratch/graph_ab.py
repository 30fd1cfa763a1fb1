"""One forward of the full model at (B, L): plain launches vs replays of one captured hipGraph (esmdiff_debug_graph_ab)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd.config import ESM3_OPEN
from esmdiff_amd.engine import Engine, _ptr
from esmdiff_amd.weights import random_init_state_dict
dev = torch.device("cuda", 0)
sd = random_init_state_dict(ESM3_OPEN, seed=0, device="cuda:0", with_geom=True)
for B, L in ((4, 60), (8, 60), (2, 258), (16, 60)):
    eng = Engine(ESM3_OPEN, sd, max_batch=B, max_len=L, device=0)
    seq = torch.randint(4, 24, (B, L), device=dev); seq[:, 0] = 0; seq[:, -1] = 2
    x = torch.full((B, L), 4096, dtype=torch.int64, device=dev); x[:, 0] = 4098; x[:, -1] = 4097
    a, b = ctypes.c_float(0), ctypes.c_float(0)
    eng._chk(eng._lib.esmdiff_debug_graph_ab(eng._h, _ptr(seq), _ptr(x), B, L, 30, ctypes.byref(a), ctypes.byref(b)))
    print(f"B={B} L={L}: direct {a.value:.3f} ms/forward, graph {b.value:.3f} ms/forward", flush=True)
    del eng
