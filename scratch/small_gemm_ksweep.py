"""Fixed vs per-K-tile cost of the small-M GEMM: M=240, N=4608, K swept (hot weights).  Run under rocprofv3; durations per K
are told apart by launch order (each K launched 50 times in a row) -> prints from the rocpd db itself if given."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
M = int(os.environ.get("M", 240)); Nn = int(os.environ.get("NN", 4608))
g = torch.Generator(device="cuda").manual_seed(0)
for K in (128, 256, 512, 1024, 1536, 3072):
    A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    for i in range(50):
        gemm_bf16(A, W, N.EPI_BF16, out=out)
    torch.cuda.synchronize()
