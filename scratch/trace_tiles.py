import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
M, Nn, K = 25800, int(os.environ.get("NN", 8192)), int(os.environ.get("KK", 1536))
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
trace = torch.zeros(4096, dtype=torch.int64, device="cuda")
for _ in range(3):
    trace.zero_()
    gemm_bf16(A, W, N.EPI_BF16, out=out, bias=trace.view(torch.float32))
torch.cuda.synchronize()
t = trace.cpu()[2048:2048 + 8 * 8 * 8].view(8, 8, 8)
names = ["setup+prologue", "K loop", "final barrier", "epilogue", "->next tile"]
for w in (0, 3, 4, 7):
    for ti in range(3):
        s = t[w, ti]
        if int(s[0]) == 0: continue
        d = [int(s[i + 1] - s[i]) for i in range(4)]
        nxt = int(t[w, ti + 1, 0] - s[4]) if int(t[w, ti + 1, 0]) else -1
        print(f"wave {w} tile {ti}: " + "  ".join(f"{n}={v}" for n, v in zip(names, d + [nxt])), " total", int(s[4] - s[0]))
