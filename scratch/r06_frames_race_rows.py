"""r06: WHICH token rows does the two-queue forward with frames get wrong (ONE block, so that a wrong (row, head) of the geometric
branch stays in its row: everything behind it — FFN, head — is row-wise)?  Run on an A/B build whose geom_attention_kernel requests
only the LDS it needs (scratch/build_variant.py geom geomlds0 -DED_GEOM_MIN_LDS_KB=0).  A wrong GEMM tile piece would show as runs
of consecutive rows (8-row DMA pieces, 32-row fragments); a wrong geometric workgroup as rows of one sample."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esmdiff_amd.config import ESM3_OPEN, ModelConfig
from esmdiff_amd.engine import Engine
from esmdiff_amd.geometry import build_affine3d_from_coordinates
from esmdiff_amd.weights import random_init_state_dict
n_layers = int(os.environ.get("LAYERS", "1"))
cfg = ESM3_OPEN if n_layers == 48 else ModelConfig(n_layers=n_layers)
FRAMES = os.environ.get("FRAMES", "1") == "1"
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
for prec in os.environ.get("PRECS", "bf16,f32_split").split(","):
    eng = Engine(cfg, sd, max_batch=B, max_len=L, precision=prec)
    if FRAMES:
        eng.set_frames(*frames)
    torch.cuda.synchronize()
    eng.set_profiling(1); ref = eng.forward_logits(x, seq, None).clone(); eng.set_profiling(0)
    for it in range(int(os.environ.get("FORWARDS", "8"))):
        o = eng.forward_logits(x, seq, None).clone()
        torch.cuda.synchronize()
        d = (o != ref)
        rows = torch.nonzero(d.any(-1))            # (sample, row)
        if rows.numel() == 0:
            n_equal = locals().get("n_equal", 0) + 1
            continue
        by = {}
        for s, r in rows.tolist():
            by.setdefault(s, []).append(r)
        desc = []
        for s, rr in by.items():
            cols = int(d[s, rr].sum(-1).float().mean())
            mx = float((o[s, rr].float() - ref[s, rr].float()).abs().max())
            desc.append(f"sample {s}: {len(rr)} rows {rr[:12]}{'…' if len(rr) > 12 else ''} (mean {cols} of {o.shape[-1]} logits differ, largest |diff| {mx:.3g}, flat rows from {s * L + rr[0]})")
        print(prec, it, "; ".join(desc), flush=True)
    print(prec, f"layers {n_layers} frames {FRAMES}: two-queue forwards bitwise equal to the one-queue forward: {locals().get('n_equal', 0)} of {int(os.environ.get('FORWARDS', '8'))}", flush=True)
    n_equal = 0
    eng.set_frames(None)
    eng.close()
