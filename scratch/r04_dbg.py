import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd.config import ModelConfig
from esmdiff_amd.engine import Engine
from esmdiff_amd.schedule import ddpm_schedule
from esmdiff_amd.weights import random_init_state_dict
def P(*a):
    print(*a, flush=True)
nl = int(sys.argv[1]); mb = int(sys.argv[2]); B = int(sys.argv[3])
g = torch.Generator().manual_seed(2)
cfg = ModelConfig(n_layers=nl)
sd = random_init_state_dict(cfg, seed=5)
L = 258
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1)
x = torch.randint(0, 4096, (B, L), generator=g)
sch = ddpm_schedule(25)
tf = sch.t_freq[12]
es = Engine(cfg, sd, max_batch=mb, max_len=L, precision="f32_split")
torch.cuda.synchronize(); P("created")
ls = es.forward_logits(x.cuda(), seq.cuda(), tf)
torch.cuda.synchronize(); P("forward ok", float(ls.abs().max()))
h = es.embeddings(B, L)
torch.cuda.synchronize(); P("emb ok")
es.close()
