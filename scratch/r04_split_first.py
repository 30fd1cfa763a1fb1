"""r04 first GPU session: (A) split-f16 GEMM numerics + timing, (B) F32_SPLIT engine vs the exact-f32 engine, (C) where the
bf16 engine's logit error comes from (body vs head).  Writes gpurun_out/r04_split_first.json."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd import _native as N  # noqa: E402
from esmdiff_amd.config import ModelConfig  # noqa: E402
from esmdiff_amd.engine import Engine, gemm_f32, gemm_split, split_rows, split_weight  # noqa: E402
from esmdiff_amd.schedule import ddpm_schedule  # noqa: E402
from esmdiff_amd.weights import random_init_state_dict  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "gpurun_out" / "r04_split_first.json"
res = {}


def rec(k, v):
    res[k] = v
    print(k, json.dumps(v), flush=True)
    OUT.parent.mkdir(exist_ok=True)
    OUT.write_text(json.dumps(res, indent=1))


def ev_time(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


# ---------------------------------------------------------------------------------------------- A
def part_a():
    g = torch.Generator().manual_seed(0)
    for (M, Nn, K, scale_a) in [(300, 1536, 1536, 1.0), (517, 4608, 1536, 1.0), (300, 1536, 4096, 0.05), (1000, 4101, 1536, 1.0),
                                (300, 1536, 1536, 1e-4), (300, 1536, 1536, 3e3)]:
        A = torch.randn(M, K, generator=g) * scale_a
        A[:, ::7] *= 1e-3                       # a spread of magnitudes inside a row (lo parts become f16 subnormals)
        W = (torch.rand(Nn, K, generator=g) * 2 - 1) / K ** 0.5
        ref = A.double() @ W.double().T
        mag = A.double().abs() @ W.double().abs().T
        a2, rs = split_rows(A.cuda())
        w2, inv = split_weight(W.cuda())
        got = gemm_split(a2, rs, w2, inv, Nn).cpu()
        e_split = float(((got.double() - ref).abs() / mag).max())
        got32 = gemm_f32(A.cuda(), W.cuda()).cpu()
        e_f32 = float(((got32.double() - ref).abs() / mag).max())
        # rms relative to the output's own rms
        r_split = float((got.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        r_f32 = float((got32.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        # split representation error alone
        hi, lo = a2[:, :K].float().cpu(), a2[:, K:2 * K].float().cpu()
        rep = float(((hi.double() + lo.double()) * rs.cpu().double()[:, None] - A.double()).abs().max() / A.abs().max())
        rec(f"gemm_{M}x{Nn}x{K}_s{scale_a}", {"split_max_rel_to_sum_abs": e_split, "f32_max_rel_to_sum_abs": e_f32,
                                                "split_rms_rel": r_split, "f32_rms_rel": r_f32, "a_repr_err_rel_rowmax": rep,
                                                "w_inv_scale": inv})
        # row independence
        sub2, srs = split_rows(A[3:4].contiguous().cuda())
        sub = gemm_split(sub2, srs, w2, inv, Nn).cpu()
        assert torch.equal(sub[0], got[3]), "row result depends on the batch"
        # residual epilogue + bias epilogue
        x0 = torch.randn(M, Nn, generator=g)
        x = x0.clone().cuda()
        if Nn % 256 == 0:
            gemm_split(a2, rs, w2, inv, Nn, N.F32EPI_RESID_DIV, out=x, div=1.1547005)
            want = x0.double() + ref / 1.1547005
            assert bool(((x.cpu().double() - want).abs() <= 2e-6 * mag + 1e-6).all()), "resid epilogue"
        bias = torch.zeros(w2.shape[0])
        bias[:Nn] = torch.randn(Nn, generator=g)
        gb = gemm_split(a2, rs, w2, inv, Nn, bias=bias.cuda()).cpu()
        assert bool(((gb.double() - (ref + bias[:Nn].double())).abs() <= 2e-6 * mag + 1e-6).all()), "bias epilogue"
    # timing at configs[1]'s launch sizes
    M = 25800
    for name, Nn, K in [("qkv", 4608, 1536), ("out", 1536, 1536), ("ffn_up", 8192, 1536), ("ffn_down", 1536, 4096)]:
        A = torch.randn(M, K, device="cuda")
        W = (torch.rand(Nn, K, device="cuda") * 2 - 1) / K ** 0.5
        a2, rs = split_rows(A)
        w2, inv = split_weight(W)
        out = torch.zeros(M, Nn, device="cuda")
        ms = ev_time(lambda: gemm_split(a2, rs, w2, inv, Nn, out=out))
        ms32 = ev_time(lambda: gemm_f32(A, W, out=out), iters=3)
        ms_split_rows = ev_time(lambda: split_rows(A))
        fl = 2.0 * M * Nn * K
        rec(f"time_{name}", {"split_ms": ms, "split_eff_tflops": fl / ms / 1e9, "mfma_tflops": 3 * fl / ms / 1e9,
                             "f32_ms": ms32, "f32_tflops": fl / ms32 / 1e9, "split_rows_ms": ms_split_rows})
        del A, W, a2, rs, w2, out


# ---------------------------------------------------------------------------------------------- B
def _seq(B, L, g):
    return torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)


def part_b():
    g = torch.Generator().manual_seed(1)
    cfg = ModelConfig(n_layers=3)
    sd = random_init_state_dict(cfg, seed=3)
    B, L = 3, 258
    seq = _seq(B, L, g)
    x = torch.randint(0, 4096, (B, L), generator=g)
    x[torch.rand(B, L, generator=g) < 0.5] = 4096
    sch = ddpm_schedule(25)
    e32 = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32")
    l32 = e32.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[5]).clone()
    h32 = e32.embeddings(B, L)
    e32.close()
    es = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
    ls = es.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[5]).clone()
    hs = es.embeddings(B, L)
    # batch independence of the split engine
    l1 = es.forward_logits(x[1:2].cuda(), seq[1:2].cuda(), sch.t_freq[5]).clone()
    es.close()
    rec("engine3_split_vs_f32", {"max_logit_diff": float((ls - l32).abs().max()), "mean": float((ls - l32).abs().mean()),
                                 "hidden_max_diff": float((hs - h32).abs().max()), "hidden_std": float(h32.std()),
                                 "row_independent": bool(torch.equal(l1[0], ls[1]))})


# ---------------------------------------------------------------------------------------------- C
def head_f64(sd, h):
    """final LayerNorm (no bias) -> Linear + b -> GELU -> LayerNorm -> Linear + b in float64 on the GPU."""
    F = torch.nn.functional
    d = lambda k: sd["net." + k].double().cuda()  # noqa: E731
    h = h.double()
    D = h.shape[-1]
    y = F.layer_norm(h, (D,), d("transformer.norm.weight"), None, 1e-5)
    y = F.linear(y, d("output_heads.structure_head.0.weight"), d("output_heads.structure_head.0.bias"))
    y = F.gelu(y)
    y = F.layer_norm(y, (D,), d("output_heads.structure_head.2.weight"), d("output_heads.structure_head.2.bias"), 1e-5)
    return F.linear(y, d("output_heads.structure_head.3.weight"), d("output_heads.structure_head.3.bias"))


def part_c(n_layers=48):
    g = torch.Generator().manual_seed(2)
    cfg = ModelConfig(n_layers=n_layers)
    t0 = time.time()
    sd = random_init_state_dict(cfg, seed=5)
    print("state dict", time.time() - t0, flush=True)
    B, L = 8, 258
    seq = _seq(B, L, g)
    x = torch.randint(0, 4096, (B, L), generator=g)
    x[torch.rand(B, L, generator=g) < 0.5] = 4096
    sch = ddpm_schedule(25)
    tf = sch.t_freq[12]
    es = Engine(cfg, sd, max_batch=100, max_len=L, precision="f32_split")
    ls = es.forward_logits(x.cuda(), seq.cuda(), tf).clone()
    hs = es.embeddings(B, L)
    # timing of the split engine at configs[1]'s batch
    xb = x[:1].repeat(100, 1).cuda()
    sb = seq[:1].repeat(100, 1).cuda()
    ms = ev_time(lambda: es.forward_logits(xb, sb, tf), iters=2)
    es.set_profiling(1)
    es.forward_logits(xb, sb, tf)
    prof = es.get_profile()
    es.set_profiling(0)
    rec(f"split_engine_forward_B100_L258_{n_layers}blocks", {"ms": ms, "samples_per_s_26_forwards": 100 / (26 * ms / 1e3),
                                                            "sections_ms": {k: round(v["ms"], 3) for k, v in prof.items()}})
    es.close()
    eb = Engine(cfg, sd, max_batch=B, max_len=L, precision="bf16")
    lb = eb.forward_logits(x.cuda(), seq.cuda(), tf).clone()
    hb = eb.embeddings(B, L)
    eb.close()
    mask = (x == 4096).cuda()
    ref = head_f64(sd, hs)          # float64 head on the split engine's hidden state ~ the float32 chain's logits
    via = head_f64(sd, hb)          # exact head on the bf16 engine's hidden state: the BODY's share of the error
    st = lambda d: {"max": float(d.abs().max()), "mean": float(d.abs().mean()), "masked_rows_max": float(d[mask].abs().max())}  # noqa: E731
    rec(f"decomposition_{n_layers}blocks", {
        "split_engine_logits_vs_f64_head_of_its_hidden": st(ls.double() - ref),
        "total_bf16_vs_ref": st(lb.double() - ref),
        "body_only__exact_head_on_bf16_hidden_vs_ref": st(via - ref),
        "head_only__bf16_logits_vs_exact_head_on_bf16_hidden": st(lb.double() - via),
        "hidden_diff": {"max": float((hb - hs).abs().max()), "mean": float((hb - hs).abs().mean()), "std": float(hs.std())},
        "logit_std": float(ref.std())})
    if n_layers == 48 and "--f32ref" in sys.argv:
        e32 = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32")
        l32 = e32.forward_logits(x.cuda(), seq.cuda(), tf).clone()
        e32.close()
        rec("split_vs_f32_48blocks", {"max": float((ls - l32).abs().max()), "mean": float((ls - l32).abs().mean())})


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "abc"
    if "a" in which:
        part_a()
    if "b" in which:
        part_b()
    if "c" in which:
        part_c()
