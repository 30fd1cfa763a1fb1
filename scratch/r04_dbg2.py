import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_split, split_rows, split_weight
def P(*a):
    print(*a, flush=True)
for M in (2064, 25800):
    for name, Nn, K in [("out", 1536, 1536), ("down", 1536, 4096), ("qkv", 4608, 1536), ("head", 4101, 1536)]:
        A = torch.randn(M, K, device="cuda")
        W = (torch.rand(Nn, K, device="cuda") * 2 - 1) / K ** 0.5
        a2, rs = split_rows(A)
        w2, inv = split_weight(W)
        ref = (A.double() @ W.double().T)
        if Nn % 256 == 0:
            x = torch.zeros(M, Nn, device="cuda")
            gemm_split(a2, rs, w2, inv, Nn, N.F32EPI_RESID_DIV, out=x, div=2.0)
            torch.cuda.synchronize()
            P(M, name, "resid ok", float((x.double() * 2 - ref).abs().max()))
        bias = torch.zeros(w2.shape[0], device="cuda")
        y = gemm_split(a2, rs, w2, inv, Nn, bias=bias)
        torch.cuda.synchronize()
        P(M, name, "store+bias ok", float((y.double() - ref).abs().max()))
        y = gemm_split(a2, rs, w2, inv, Nn)
        torch.cuda.synchronize()
        P(M, name, "store ok", float((y.double() - ref).abs().max()))
