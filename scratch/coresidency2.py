"""Does a light memory-bound kernel (torch elementwise, < 32 VGPRs) run UNDER the four-wave GEMM of another stream?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
M = int(os.environ.get("MM", 12900))
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, 1536, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
Wt = ((torch.rand(8192, 1536, generator=g, device="cuda") * 2 - 1) / 39).to(torch.bfloat16)
out = torch.empty(M, 4096, dtype=torch.bfloat16, device="cuda")
x = torch.randn(M, 1536 * 4, device="cuda")          # 317 MB read + write per pass
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(do_gemm, do_ew, n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); s1.wait_event(a); s2.wait_event(a)
    for _ in range(n):
        if do_gemm:
            with torch.cuda.stream(s1):
                gemm_bf16(A, Wt, N.EPI_SWIGLU_BF16, out=out)
        if do_ew:
            with torch.cuda.stream(s2):
                x.mul_(1.0001)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s1); e2.record(s2)
    torch.cuda.current_stream().wait_event(e1); torch.cuda.current_stream().wait_event(e2)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
run(True, True, 3)
print(f"GEMM alone {run(True, False):.1f} us   elementwise alone {run(False, True):.1f} us   both {run(True, True):.1f} us")
