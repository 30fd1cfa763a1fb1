"""r04: the split attention inside the F32_SPLIT engine: logits vs the exact-f32 engine (3 blocks), ragged shapes, forward time
at configs[1]'s batch with the per-section breakdown."""
import json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd.config import ModelConfig, TINY
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict

def seq_(B, L, g):
    return torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)

g = torch.Generator().manual_seed(1)
sch = ddpm_schedule(25)
res = {}
for cfg, shapes in ((TINY, [(1, 1), (1, 2), (2, 3), (3, 31), (2, 33), (1, 64), (2, 65), (1, 127), (2, 129), (1, 300)]),
                    (ModelConfig(n_layers=3), [(2, 60), (3, 258), (1, 1026)])):
    sd = random_init_state_dict(cfg, seed=3)
    for B, L in shapes:
        seq = seq_(B, L, g) if L >= 2 else torch.zeros(B, 1, dtype=torch.int64)
        x = torch.randint(0, 4096, (B, L), generator=g)
        x[torch.rand(B, L, generator=g) < 0.5] = 4096
        e32 = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32")
        ref = e32.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[5]).clone(); e32.close()
        es = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
        got = es.forward_logits(x.cuda(), seq.cuda(), sch.t_freq[5]).clone()
        alone = es.forward_logits(x[-1:].cuda(), seq[-1:].cuda(), sch.t_freq[5]).clone(); es.close()
        d = float((got - ref).abs().max())
        print(cfg.d_model, B, L, "max diff vs exact f32", d, "row independent", bool(torch.equal(alone[0], got[-1])), flush=True)
        res[f"d{cfg.d_model}_B{B}_L{L}"] = d
        assert d < 3e-5, d
cfg48 = ModelConfig()
sd48 = random_init_state_dict(cfg48, seed=0, device="cuda")
B, L = 100, 258
seq = seq_(B, L, g).cuda()
x = torch.full((B, L), 4096, dtype=torch.int64).cuda()
es = Engine(cfg48, sd48, max_batch=B, max_len=L, precision="f32_split")
es.forward_logits(x, seq, sch.t_freq[0]); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    es.forward_logits(x, seq, sch.t_freq[0])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
es.set_profiling(1); es.forward_logits(x, seq, sch.t_freq[0]); prof = es.get_profile(); es.set_profiling(0)
print("forward ms", ms, "samples/s", 100 / (26 * ms / 1e3), {k: round(v["ms"], 2) for k, v in prof.items()}, flush=True)
res["forward_ms_B100"] = ms; res["sections"] = {k: round(v["ms"], 2) for k, v in prof.items()}
t0 = time.perf_counter(); ids = es.ddpm_sample(seq, sch, seed=1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("ddpm_sample samples/s", 100 / dt)
res["ddpm_samples_per_s"] = 100 / dt
es.close()
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/r04_split_attn.json").write_text(json.dumps(res, indent=1))
