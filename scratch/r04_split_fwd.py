import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd.config import ModelConfig
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "f32_split"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = ModelConfig(n_layers=nl)
sd = random_init_state_dict(cfg, seed=0, device="cuda")
B, L = 100, 258
g = torch.Generator().manual_seed(1)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
x = torch.full((B, L), 4096, dtype=torch.int64).cuda()
sch = ddpm_schedule(25)
e = Engine(cfg, sd, max_batch=B, max_len=L, precision=prec)
for _ in range(4):
    e.forward_logits(x, seq, sch.t_freq[0])
torch.cuda.synchronize()
e.close()
