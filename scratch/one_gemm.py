import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
M, Nn, K = int(os.environ.get("MM", 25800)), 8192, 1536
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
Wg = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
out = torch.empty(M, Nn // 2, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    gemm_bf16(A, Wg, N.EPI_SWIGLU_BF16, out=out)
torch.cuda.synchronize()
print("done")
