"""Joules per launch of the non-GEMM kernels at configs[1] shapes: each kernel is launched back to back for ~3 s while a
thread samples the package power (amdgpu hwmon, as bench.py does).  energy = mean W x mean us per launch; the idle package
power measured first is what any kernel pays just for existing that long.  (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PowerSampler
from esmdiff_amd import _native as N
from esmdiff_amd.config import ModelConfig
from esmdiff_amd.engine import Engine, gemm_bf16, layernorm_bf16
from esmdiff_amd.weights import random_init_state_dict

B, L, D = 100, 258, 1536
M = B * L
cfg = ModelConfig(n_layers=1)
eng = Engine(cfg, random_init_state_dict(cfg, seed=0, device="cuda:0"), max_batch=B, max_len=L)
qkv = torch.randn(M, 3 * D, device="cuda").to(torch.bfloat16)
ones = torch.ones(D, device="cuda")
x = torch.randn(M, D, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, D, generator=g, device="cuda").to(torch.bfloat16)
Wup = (torch.randn(8192, D, generator=g, device="cuda") / D ** 0.5).to(torch.bfloat16)
out_up = torch.empty(M, 4096, dtype=torch.bfloat16, device="cuda")

def measure(name, fn, secs=3.0, per_call=1):
    fn(); torch.cuda.synchronize()
    ps = PowerSampler(0); ps.start()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.perf_counter()
    ev0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    ev1.record(); torch.cuda.synchronize()
    rec = ps.stop()
    us = ev0.elapsed_time(ev1) * 1e3 / (n * per_call)
    w = rec["mean_w"] if rec else float("nan")
    print(f"{name:34s} {us:8.1f} us/launch  {w:7.1f} W  {w * us * 1e-6:7.4f} J/launch  sclk {rec['mean_sclk_mhz'] if rec else 0:.0f} MHz", flush=True)

ps = PowerSampler(0); ps.start(); time.sleep(2.0); print("idle:", ps.stop())
measure("layernorm (f32 -> bf16)", lambda: layernorm_bf16(x, ones, ones))
measure("qk_norm_rope + attention", lambda: eng.attention(qkv, ones, ones, B, L))
measure("FFN-up GEMM (SwiGLU) M=25800", lambda: gemm_bf16(A, Wup, N.EPI_SWIGLU_BF16, out=out_up))
