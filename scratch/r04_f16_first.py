"""r04: first run of the f16 engine (precision="f16"): logits vs the F32_SPLIT engine next to the bf16 engine's, small-batch path,
two streams, coordinates; forward time at configs[1]'s batch."""
import json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd.config import ModelConfig
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict

def seq_(B, L, g):
    return torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)

res = {}
g = torch.Generator().manual_seed(1)
cfg = ModelConfig(n_layers=3)
sd = random_init_state_dict(cfg, seed=3, with_geom=True)
sch = ddpm_schedule(25)
for (B, L) in [(2, 60), (3, 258), (12, 258)]:
    seq = seq_(B, L, g)
    x = torch.randint(0, 4096, (B, L), generator=g)
    x[torch.rand(B, L, generator=g) < 0.5] = 4096
    ref = None
    out = {}
    for prec in ("f32_split", "bf16", "f16"):
        e = Engine(cfg, sd, max_batch=B, max_len=L, precision=prec)
        lg = e.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[5]).clone()
        hid = e.embeddings(B, L)
        ids = e.ddpm_sample(seq.cuda(), ddpm_schedule(4), seed=3).cpu()
        e.close()
        if ref is None:
            ref, href = lg, hid
        else:
            d = (lg - ref).abs()
            out[prec] = {"max": float(d.max()), "mean": float(d.mean()), "hidden_mean": float((hid - href).abs().mean()),
                         "masks_left": int((ids == 4096).sum())}
    res[f"B{B}_L{L}"] = out
    print(B, L, json.dumps(out), flush=True)
# f32 head on the f16 body
B, L = 3, 258
seq = seq_(B, L, g); x = torch.full((B, L), 4096)
e0 = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split"); ref = e0.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[2]).clone(); e0.close()
for kw in ({"precision": "f16"}, {"precision": "f16", "head_precision": "f32"}, {"precision": "bf16", "head_precision": "f32"}):
    e = Engine(cfg, sd, max_batch=B, max_len=L, **kw)
    lg = e.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[2]).clone(); e.close()
    d = (lg - ref).abs(); print(kw, float(d.max()), float(d.mean()), flush=True)
    res[str(kw)] = {"max": float(d.max()), "mean": float(d.mean())}
# speed at configs[1]
cfg48 = ModelConfig()
sd48 = random_init_state_dict(cfg48, seed=0, device="cuda")
B, L = 100, 258
seq = seq_(B, L, g).cuda()
sch = ddpm_schedule(25)
for prec in ("bf16", "f16", "bf16", "f16"):
    e = Engine(cfg48, sd48, max_batch=B, max_len=L, precision=prec)
    e.ddpm_sample(seq, sch, seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2):
        e.ddpm_sample(seq, sch, seed=k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(prec, "samples/s", B / dt, flush=True)
    res.setdefault("speed", {}).setdefault(prec, []).append(B / dt)
    e.close()
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/r04_f16_first.json").write_text(json.dumps(res, indent=1))
