import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd.engine import gemm_split, split_rows, split_weight
M, Nn, K = int(os.environ.get("MM", 25800)), 8192, 1536
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.rand(M, K, generator=g, device="cuda") * 2 - 1
W = (torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5
a3, rs = split_rows(A)
w3, inv = split_weight(W)
out = torch.empty(M, Nn, dtype=torch.float32, device="cuda")
for _ in range(3):
    gemm_split(a3, rs, w3, inv, Nn, out=out)
torch.cuda.synchronize()
print("done")
