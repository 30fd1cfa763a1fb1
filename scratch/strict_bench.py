"""Timing of the float32 (strict) path: the f32 MFMA GEMM alone at the decoder's shapes, and a whole 100-sample decode /
a B = 100 strict forward of the 48-block model.  python scratch/strict_bench.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from esmdiff_amd.config import DecoderConfig, ESM3_OPEN
from esmdiff_amd.engine import StructureDecoder, Engine, gemm_f32
from esmdiff_amd.weights import random_init_decoder_state_dict, random_init_state_dict

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

for M, N, K in ((25800, 3840, 1280), (25800, 7168, 1280), (25800, 1280, 3584), (25800, 4608, 1536), (6400, 3840, 1280), (240, 4608, 1536)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    out = torch.empty(M, N, device="cuda")
    dt = timeit(lambda: gemm_f32(A, W, out=out))
    print(f"gemm_f32 M={M} N={N} K={K}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.1f} TF/s")

cfg = DecoderConfig()
sd = random_init_decoder_state_dict(cfg, seed=1, device="cuda")
tok = torch.randint(0, 4096, (100, 258), device="cuda"); tok[:, 0], tok[:, -1] = 4098, 4097
for prec in ("f32", "bf16"):
    dec = StructureDecoder(cfg, sd, max_batch=100, max_len=258, precision=prec)
    dt = timeit(lambda: dec.decode(tok, return_plddt=True), n=3)
    print(f"decoder {prec}: 100 x 258 tokens decode {dt*1e3:.1f} ms")
    dec.close()
del sd
sd = random_init_state_dict(ESM3_OPEN, seed=1, device="cuda")
from esmdiff_amd.schedule import ddpm_schedule
sch = ddpm_schedule(25)
for prec, B in (("f32", 100), ("f32", 4), ("bf16", 100)):
    eng = Engine(ESM3_OPEN, sd, max_batch=B, max_len=258, precision=prec)
    x = torch.full((B, 258), 4096, dtype=torch.int64, device="cuda")
    seq = torch.randint(4, 24, (B, 258), device="cuda"); seq[:, 0], seq[:, -1] = 0, 2
    dt = timeit(lambda: eng.forward_logits(x, seq, sch.t_freq[0]), n=3)
    print(f"ESM3-open forward {prec} B={B} L=258: {dt*1e3:.1f} ms")
    if prec == "f32" and B == 100:
        eng.set_profiling(1); eng.forward_logits(x, seq, sch.t_freq[0]); print({k: round(v['ms'], 2) for k, v in eng.get_profile().items()}); eng.set_profiling(0)
    eng.close()
