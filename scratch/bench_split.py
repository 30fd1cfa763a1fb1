import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16_timed
g = torch.Generator(device="cuda").manual_seed(0)
for (M, Nn, K) in [(25800, 1536, 1536), (21760, 1536, 1536), (21504, 1536, 1536), (4040, 1536, 1536), (25800, 1536, 4096), (21760, 1536, 4096), (4040, 1536, 4096), (16384,1536,1536),(10752,1536,1536)]:
    A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
    ms = min(gemm_bf16_timed(A, W, out, N.EPI_BF16, iters=20) for _ in range(3))
    print(f"M={M} N={Nn} K={K}: {ms*1e3:8.1f} us {2.0*M*Nn*K/ms/1e9:8.1f} TF/s  tiles256={((M+255)//256)*(Nn//256)}")
