import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16_timed
M = 25800
g = torch.Generator(device="cuda").manual_seed(0)
for Nn, K in [(1536, 4096), (1536, 4160), (1536, 4032), (8192, 1536), (8192, 1600), (8192, 1472), (4608, 1536), (4608, 2048), (4608, 2112)]:
    A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
    ms = min(gemm_bf16_timed(A, W, out, N.EPI_BF16, iters=10) for _ in range(3))
    print(f"N={Nn} K={K} (row stride {K*2} B): {ms*1e3:8.1f} us {2.0*M*Nn*K/ms/1e9:8.1f} TF/s")
