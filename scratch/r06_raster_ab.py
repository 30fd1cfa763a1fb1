"""r06 A/B of the 256x256 GEMM's tile raster: per variant library (ESMDIFF_LIB=...) the four block linears
at M = 12 900 alone on the GPU (HIP events over 30 launches, correctness of a slice against a float32 matmul), then the whole
sampling job (bf16 engine, 100 x 258 tokens, 25 steps: 1 warm + 2 timed) with package power and clock beside it.
    ESMDIFF_LIB=esmdiff_amd/lib/libesmdiff_hip_<tag>.so python scratch/r06_raster_ab.py
The fabric traffic of the same variant comes from tools/pmc_traffic.py (separate rocprofv3 --pmc passes)."""
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import Engine, gemm_bf16
from bench import PowerSampler

M = int(os.environ.get("M", 12900))
rec = {"lib": os.path.basename(str(N.lib_path())), "M": M, "gemm_us": {}}
g = torch.Generator(device="cuda").manual_seed(0)
for name, Nn, K, epi in (("qkv", 4608, 1536, N.EPI_BF16), ("out", 1536, 1536, N.EPI_BF16), ("ffn_up", 8192, 1536, N.EPI_SWIGLU_BF16),
                         ("ffn_down", 1536, 4096, N.EPI_BF16)):
    A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, Nn // 2 if epi == N.EPI_SWIGLU_BF16 else Nn, dtype=torch.bfloat16, device="cuda")
    for _ in range(5):
        gemm_bf16(A, W, epi, out=out)
    if epi == N.EPI_BF16:
        err = max(float((out[:300].float() - A[:300].float() @ W.float().t()).abs().max()),
                  float((out[-300:].float() - A[-300:].float() @ W.float().t()).abs().max()))
        assert err < 0.05, (name, err)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(30):
            gemm_bf16(A, W, epi, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30)
    rec["gemm_us"][name] = round(best * 1e3, 1)
    rec.setdefault("gemm_tflops", {})[name] = round(2.0 * M * Nn * K / best / 1e9, 1)
del A, W, out
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict
B, L, T = 100, 258, 25
sd = random_init_state_dict(cfg, seed=0, device="cuda")
eng = Engine(cfg, sd, max_batch=B, max_len=L)
del sd
gg = torch.Generator().manual_seed(258)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=gg), torch.tensor([2])])[None].repeat(B, 1).cuda()
sch = ddpm_schedule(T, freq_dim=cfg.freq_dim)
ids0 = eng.ddpm_sample(seq, sch, seed=1)
ps = PowerSampler(0); ps.start()
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(2):
    ids = eng.ddpm_sample(seq, sch, seed=1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
rec["power"] = ps.stop()
rec["samples_per_s"] = round(B / dt, 2)
rec["ids_checksum"] = int(ids.sum())
rec["ids_repeatable"] = bool(torch.equal(ids, ids0))
print(json.dumps(rec))
