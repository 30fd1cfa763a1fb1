"""r06: do two 256x256 GEMMs on two HIP streams disturb each other?  Shapes of block 0's geometric branch (N = 3840 / K = 768)
against a block linear (control), each alone first (reference), then 20 rounds of both streams at once."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
g = torch.Generator(device="cuda").manual_seed(0)
M = 12900
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, Nn, K in (("control out-proj", 1536, 1536), ("geom proj", 3840, 1536), ("geom out", 1536, 768), ("qkv", 4608, 1536)):
    A1 = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    A2 = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    r1 = gemm_bf16(A1, W, N.EPI_BF16).clone(); r2 = gemm_bf16(A2, W, N.EPI_BF16).clone()
    torch.cuda.synchronize()
    o1, o2 = torch.empty_like(r1), torch.empty_like(r2)
    bad = []
    for it in range(20):
        o1.zero_(); o2.zero_(); torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            gemm_bf16(A1, W, N.EPI_BF16, out=o1)
        with torch.cuda.stream(s2):
            gemm_bf16(A2, W, N.EPI_BF16, out=o2)
        torch.cuda.synchronize()
        for tag, o, r in (("s1", o1, r1), ("s2", o2, r2)):
            d = (o != r).any(1)
            if bool(d.any()):
                rows = torch.nonzero(d).flatten()
                bad.append((it, tag, int(d.sum()), int(rows.min()), int(rows.max())))
    print(f"{name:18s} N={Nn} K={K}: rounds with wrong rows {len(bad)} of 40; first {bad[:4]}", flush=True)
