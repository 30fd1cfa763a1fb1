import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
M, Nn, K = 25800, 8192, 1536
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
trace = torch.zeros(8 * 32 * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    gemm_bf16(A, W, N.EPI_BF16, out=out, bias=trace.view(torch.float32))
torch.cuda.synchronize()
t = trace.cpu().view(8, 32, 8)
names = ["g1-g4", "g5-g7", "vmcnt", "barrier", "g8 issue", "g8 mma", "->next"]
print("per wave, K-tiles 6..11 (cycles, 100MHz? memtime ticks):")
for w in range(8):
    for kt in (6, 7, 8):
        s = t[w, kt]
        d = [int(s[i + 1] - s[i]) for i in range(6)] + [int(t[w, kt + 1, 0] - s[6])]
        print(f"wave {w} kt {kt}: " + "  ".join(f"{n}={v}" for n, v in zip(names, d)), " total", int(t[w, kt + 1, 0] - s[0]))
print("skew at barrier arrival (stamp3) kt=8:", [int(t[w, 8, 3] - t[0, 8, 3]) for w in range(8)])
