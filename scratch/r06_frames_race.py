"""r06: where does the two-queue forward with frames become non-deterministic?  (r06 records: run on builds whose geom_attention_kernel requested 12.5 / 16 / 40 KB of LDS; the product now requests 40 KB.)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esmdiff_amd.config import ESM3_OPEN, ModelConfig
from esmdiff_amd.engine import Engine
from esmdiff_amd.geometry import build_affine3d_from_coordinates
from esmdiff_amd.weights import random_init_state_dict
n_layers = int(os.environ.get("LAYERS", "48"))
cfg = ESM3_OPEN if n_layers == 48 else ModelConfig(n_layers=n_layers)
sd = random_init_state_dict(cfg, seed=11, device="cuda", with_geom=True)
B, L = 100, 258
g = torch.Generator().manual_seed(1)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
x = torch.randint(0, 4096, (B, L), generator=g); x[:, 0], x[:, -1] = 4098, 4097; x[:, 97:161] = 4096
x = x.cuda()
ca = torch.cumsum(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1) * 3.8, 0)
xyz = torch.stack([ca + torch.tensor([-1.2, 0.7, 0.0]), ca, ca + torch.tensor([1.3, 0.6, 0.1])], 1)
xyz[97:161] = float("inf"); xyz[0] = xyz[-1] = float("nan")
frames = tuple(f.cuda() for f in build_affine3d_from_coordinates(xyz[None].repeat(B, 1, 1, 1)))
for prec in os.environ.get("PRECS", "f32_split,f16").split(","):
    eng = Engine(cfg, sd, max_batch=B, max_len=L, precision=prec)
    def diff(a, b):
        return torch.nonzero((a != b).flatten(1).any(1)).flatten().tolist()
    # (1) frames set once, several forwards
    eng.set_frames(*frames); torch.cuda.synchronize()
    outs = [eng.forward_logits(x, seq, None).clone() for _ in range(4)]
    print(prec, n_layers, "frames set once, 4 forwards: differing vs first", [diff(outs[0], o) for o in outs[1:]], flush=True)
    # (2) the same with a device synchronisation between the forwards
    outs = []
    for _ in range(4):
        outs.append(eng.forward_logits(x, seq, None).clone()); torch.cuda.synchronize()
    print(prec, n_layers, "… with synchronize between forwards:", [diff(outs[0], o) for o in outs[1:]], flush=True)
    # (3) one queue (profiling mode 1) as the reference
    eng.set_profiling(1); ref = eng.forward_logits(x, seq, None).clone(); eng.set_profiling(0)
    print(prec, n_layers, "two queues vs one queue:", [diff(ref, o) for o in outs], flush=True)
    eng.set_frames(None)
    eng.close()
