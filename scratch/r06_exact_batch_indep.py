"""r06 diagnostics: is the F32_SPLIT engine's forward bitwise independent of the batch a sample sits in — with and without frames?"""
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
x = torch.randint(0, 4096, (B, L), generator=g)
x[:, 0], x[:, -1] = 4098, 4097
x[:, 97:161] = 4096
x[:, 97:161][torch.rand(B, 64, generator=g) < 0.5] = 7
x = x.cuda()
ca = torch.cumsum(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1) * 3.8, 0)
xyz = torch.stack([ca + torch.tensor([-1.2, 0.7, 0.0]), ca, ca + torch.tensor([1.3, 0.6, 0.1])], 1)
xyz[97:161] = float("inf"); xyz[0] = xyz[-1] = float("nan")
frames = build_affine3d_from_coordinates(xyz[None].repeat(B, 1, 1, 1))
for prec in ("f32_split",):
    eng = Engine(cfg, sd, max_batch=B, max_len=L, precision=prec)
    for with_frames in (False, True):
        def fwd(idx):
            if with_frames:
                eng.set_frames(*(f[idx] for f in frames))
            out = eng.forward_logits(x[idx], seq[idx], None).clone()
            if with_frames:
                eng.set_frames(None)
            return out
        full = fwd(torch.arange(B))
        for name, idx in (("one", torch.tensor([3])), ("first40", torch.arange(40)), ("mid 30..69", torch.arange(30, 70)), ("33 scattered", torch.arange(0, 99, 3)),
                          ("64 rev", torch.arange(63, -1, -1))):
            sub = fwd(idx)
            d = (sub - full[idx.cuda()]).abs()
            print(f"{prec} frames={with_frames} subset {name:12s}: rows {len(idx)} max |diff| {float(d.max()):.3e} differing samples {int((d.amax((1, 2)) > 0).sum())}", flush=True)
    eng.close()
eng = Engine(cfg, sd, max_batch=B, max_len=L, precision="f32_split")
def fwd(idx, prof=0):
    eng.set_profiling(prof)
    eng.set_frames(*(f[idx] for f in frames))
    out = eng.forward_logits(x[idx], seq[idx], None).clone()
    eng.set_frames(None)
    eng.set_profiling(0)
    return out
allidx = torch.arange(B)
a, b2, c1 = fwd(allidx), fwd(allidx), fwd(allidx, 1)
print("full twice (two streams): differing samples", torch.nonzero((a != b2).flatten(1).any(1)).flatten().tolist())
print("full two streams vs one stream: differing samples", torch.nonzero((a != c1).flatten(1).any(1)).flatten().tolist())
ones = torch.cat([fwd(torch.tensor([i])) for i in (0, 1, 48, 49, 50, 51, 98, 99)])
ref_idx = torch.tensor([0, 1, 48, 49, 50, 51, 98, 99]).cuda()
print("singles vs two-stream full:", torch.nonzero((ones != a[ref_idx]).flatten(1).any(1)).flatten().tolist(), " vs one-stream full:", torch.nonzero((ones != c1[ref_idx]).flatten(1).any(1)).flatten().tolist())
for n in (40,):
    s2, s1 = fwd(torch.arange(n)), fwd(torch.arange(n), 1)
    print(f"first {n}: two-stream vs one-stream differing", torch.nonzero((s2 != s1).flatten(1).any(1)).flatten().tolist(), "; one-stream vs full one-stream", torch.nonzero((s1 != c1[:n]).flatten(1).any(1)).flatten().tolist())
    d = (s2 - s1).abs().amax(2)
    bad = torch.nonzero((s2 != s1).flatten(1).any(1)).flatten().tolist()
    for bb in bad:
        print("  sample", bb, "rows differing", int((d[bb] > 0).sum()), "first rows", torch.nonzero(d[bb] > 0).flatten()[:6].tolist())
