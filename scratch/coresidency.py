"""Do memory-bound kernels of one stream run UNDER a GEMM of another stream (same CUs, left-over registers / LDS)?
t(GEMM alone), t(rope + attention alone), t(both launched on two streams)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd import _native as N
from esmdiff_amd.config import ModelConfig
from esmdiff_amd.engine import Engine, gemm_bf16
from esmdiff_amd.weights import random_init_state_dict
cfg = ModelConfig(n_layers=1)
B, L = 50, 258
eng = Engine(cfg, random_init_state_dict(cfg, seed=0, device="cuda:0"), max_batch=B, max_len=L)
M = B * L
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, 1536, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
Wt = ((torch.rand(8192, 1536, generator=g, device="cuda") * 2 - 1) / 39).to(torch.bfloat16)
out = torch.empty(M, 4096, dtype=torch.bfloat16, device="cuda")
qkv = torch.randn(M, 3 * 1536, device="cuda").to(torch.bfloat16)
w = torch.ones(1536, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(do_gemm, do_attn, n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    s1.wait_event(a); s2.wait_event(a)
    for _ in range(n):
        if do_gemm:
            with torch.cuda.stream(s1):
                gemm_bf16(A, Wt, N.EPI_SWIGLU_BF16, out=out)
        if do_attn:
            with torch.cuda.stream(s2):
                eng.attention(qkv, w, w, B, L)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s1); e2.record(s2)
    torch.cuda.current_stream().wait_event(e1); torch.cuda.current_stream().wait_event(e2)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for _ in range(2):
    run(True, True, 3)
print(f"GEMM (FFN-up, M={M}) alone      {run(True, False):8.1f} us")
print(f"rope + attention alone          {run(False, True):8.1f} us")
print(f"both, two streams               {run(True, True):8.1f} us   (sum would be serial; max would be perfect overlap)")
