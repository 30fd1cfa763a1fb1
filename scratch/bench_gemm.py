"""micro-benchmark of the GEMM shapes of one ESM3 block at B*L = 25800 (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16, gemm_bf16_timed

M = int(os.environ.get("M", 25800))
shapes = [("qkv", 4608, 1536, N.EPI_BF16), ("out", 1536, 1536, N.EPI_BF16), ("out_resid", 1536, 1536, N.EPI_RESID_F32),
          ("ffn_up", 8192, 1536, N.EPI_SWIGLU_BF16), ("ffn_down", 1536, 4096, N.EPI_BF16),
          ("ffn_down_resid", 1536, 4096, N.EPI_RESID_F32)]
g = torch.Generator(device="cuda").manual_seed(0)
tot_f = tot_t = 0
for name, Nn, K, epi in shapes:
    A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    if epi == N.EPI_RESID_F32:
        out = torch.zeros(M, Nn, device="cuda")
    elif epi == N.EPI_SWIGLU_BF16:
        out = torch.empty(M, Nn // 2, dtype=torch.bfloat16, device="cuda")
    else:
        out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
    # correctness on a slice
    if epi == N.EPI_BF16:
        o = gemm_bf16(A, W, epi, out=out)
        ref = A[:512].float() @ W.float().t()
        err = float((o[:512].float() - ref).abs().max())
        ref2 = A[-300:].float() @ W.float().t()
        err2 = float((o[-300:].float() - ref2).abs().max())
    else:
        err = err2 = float("nan")
    ms = min(gemm_bf16_timed(A, W, out, epi, iters=20, alpha=0.0 if epi == N.EPI_RESID_F32 else 1.0) for _ in range(3))
    fl = 2.0 * M * Nn * K
    print(f"{name:16s} M={M} N={Nn} K={K}  {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TF/s   max|err| {err:.4f} {err2:.4f}")
    if "resid" not in name:
        tot_f += fl; tot_t += ms
print(f"block linears (bf16-store epilogues): {tot_f/tot_t/1e9:.1f} TF/s  ({tot_t*1e3:.1f} us per block)")
