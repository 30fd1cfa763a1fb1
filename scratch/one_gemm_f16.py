import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_f16
M, Nn, K = int(os.environ.get("MM", 25800)), 8192, 1536
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).half()
Wg = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).half()
out = torch.empty(M, Nn // 2, dtype=torch.float16, device="cuda")
for _ in range(3):
    gemm_f16(A, Wg, N.EPI_SWIGLU_BF16, out=out)
torch.cuda.synchronize()
print("done")
