"""Small-M GEMM shapes of one block with the weights hot (same matrix every launch: served by L2 / Infinity Cache) or cold
(48 different matrices in turn, as in the forward: HBM).  Run under rocprofv3 --kernel-trace --stats; MODE=hot|cold, M=240."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16

M = int(os.environ.get("M", 240))
mode = os.environ.get("MODE", "cold")
shape = os.environ.get("SHAPE", "ffn_up")
Nn, K, epi = {"qkv": (4608, 1536, N.EPI_BF16), "ffn_up": (8192, 1536, N.EPI_SWIGLU_BF16), "out": (1536, 1536, N.EPI_BF16),
              "ffn_down": (1536, 4096, N.EPI_BF16)}[shape]
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
nW = 48 if mode == "cold" else 1
Ws = [((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16) for _ in range(nW)]
out = torch.empty(M, Nn // 2 if epi == N.EPI_SWIGLU_BF16 else Nn, dtype=torch.bfloat16, device="cuda")
torch.cuda.synchronize()
for it in range(4):
    for i in range(48):
        gemm_bf16(A, Ws[i % nW], epi, out=out)
torch.cuda.synchronize()
