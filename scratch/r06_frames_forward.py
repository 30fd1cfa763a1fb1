"""r06: three forwards of the full model with backbone frames (configs[4]'s shape: 100 x 258 tokens, residues 96..159 masked) for a kernel trace:
    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_frames -o frames --output-format rocpd -- python $R/scratch/r06_frames_forward.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esmdiff_amd.config import ESM3_OPEN as cfg
from esmdiff_amd.engine import Engine
from esmdiff_amd.geometry import build_affine3d_from_coordinates
from esmdiff_amd.weights import random_init_state_dict
sd = random_init_state_dict(cfg, seed=11, device="cuda", with_geom=True)
B, L = 100, 258
g = torch.Generator().manual_seed(1)
seq = torch.cat([torch.tensor([0]), torch.randint(4, 24, (L - 2,), generator=g), torch.tensor([2])])[None].repeat(B, 1).cuda()
x = torch.randint(0, 4096, (B, L), generator=g); x[:, 0], x[:, -1] = 4098, 4097; x[:, 97:161] = 4096
x = x.cuda()
ca = torch.cumsum(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1) * 3.8, 0)
xyz = torch.stack([ca + torch.tensor([-1.2, 0.7, 0.0]), ca, ca + torch.tensor([1.3, 0.6, 0.1])], 1)
xyz[97:161] = float("inf"); xyz[0] = xyz[-1] = float("nan")
frames = build_affine3d_from_coordinates(xyz[None].repeat(B, 1, 1, 1))
eng = Engine(cfg, sd, max_batch=B, max_len=L, precision=os.environ.get("PREC", "bf16"))
eng.set_frames(*frames)
for _ in range(4):
    eng.forward_logits(x, seq, None)
torch.cuda.synchronize()
