"""r06: the product's geom_attention_kernel (12.5 KB LDS request; scratch/ubench/geom_alone.hip) outside the engine: alone, then beside
the 256x256 GEMM on another stream, in the float32 variant (full-precision outputs).  Which (sample, row, head) outputs differ from
the solo run?"""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
from esmdiff_amd.geometry import build_affine3d_from_coordinates
B, L, VH = int(os.environ.get("NB_B", "50")), 258, 256
ROUNDS = int(os.environ.get("ROUNDS", "6"))
SLEEP = int(os.environ.get("SLEEP_CYCLES", "300000"))
g = torch.Generator().manual_seed(1)
ca = torch.cumsum(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1) * 3.8, 0)
xyz = torch.stack([ca + torch.tensor([-1.2, 0.7, 0.0]), ca, ca + torch.tensor([1.3, 0.6, 0.1])], 1)
xyz[97:161] = float("inf"); xyz[0] = xyz[-1] = float("nan")
rot, trans, has = (f.cuda() for f in build_affine3d_from_coordinates(xyz[None].repeat(B, 1, 1, 1)))
rot = rot.reshape(B * L, 9).float().contiguous(); trans = trans.reshape(B * L, 3).float().contiguous()
fmask = has.reshape(B * L).to(torch.uint8).contiguous()
gd = torch.Generator(device="cuda").manual_seed(0)
P = (torch.randn(B * L, 15 * VH, generator=gd, device="cuda") * 0.7).contiguous()
w_rot = torch.rand(VH, generator=gd, device="cuda") + 0.3
w_dist = torch.rand(VH, generator=gd, device="cuda") * 0.2 + 0.05
M = 12900
A = (torch.rand(M, 1536, generator=gd, device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(3840, 1536, generator=gd, device="cuda") * 2 - 1) / 39.0).to(torch.bfloat16)
ref_g = gemm_bf16(A, W, N.EPI_BF16).clone()
og = torch.empty_like(ref_g)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for tag in os.environ.get("LIBS", "pk,nopk").split(","):
    lib = ctypes.CDLL(os.path.join(ROOT, "scratch/ubench", "geom_alone.so" if tag == "pk" else f"geom_alone_{tag}.so"))
    fn = getattr(lib, f"ga_geom_f32_{tag}")
    fn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    def run(out, stream):
        rc = fn(P.data_ptr(), rot.data_ptr(), trans.data_ptr(), fmask.data_ptr(), w_rot.data_ptr(), w_dist.data_ptr(), out.data_ptr(), B, L, VH, stream)
        assert rc == 0, rc
    solo = torch.zeros(B * L, 3 * VH, device="cuda")
    run(solo, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    again = torch.zeros_like(solo)
    run(again, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    print(f"[{tag}] solo twice equal: {bool(torch.equal(solo.view(torch.int32), again.view(torch.int32)))}; finite {bool(torch.isfinite(solo).all())}; |out| max {float(solo.abs().max()):.3g}", flush=True)
    for it in range(ROUNDS):
        out = torch.zeros_like(solo)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(4):
                gemm_bf16(A, W, N.EPI_BF16, out=og)
        with torch.cuda.stream(s2):
            torch.cuda._sleep(SLEEP)
            run(out, s2.cuda_stream)
        with torch.cuda.stream(s1):
            for _ in range(12):
                gemm_bf16(A, W, N.EPI_BF16, out=og)
        torch.cuda.synchronize()
        d = (out.view(torch.int32) != solo.view(torch.int32)).view(B, L, VH, 3).any(-1)      # (sample, row, head)
        nb = int(d.sum())
        gem_ok = bool(torch.equal(og, ref_g))
        if nb == 0:
            print(f"[{tag}] round {it}: equal to the solo run (GEMM right: {gem_ok})", flush=True)
            continue
        idx = torch.nonzero(d)
        rows = idx[:, 1]
        quarters = torch.bincount((rows % 64) // 16, minlength=4).tolist()
        trips = torch.bincount(rows // 64, minlength=5).tolist()
        samples = torch.unique(idx[:, 0]).tolist()
        s0, r0, h0 = idx[0].tolist()
        print(f"[{tag}] round {it}: {nb} (sample, row, head) outputs differ; samples {samples[:12]}{'…' if len(samples) > 12 else ''}; by lane quarter {quarters}; by query trip {trips}; "
              f"heads hit {int(torch.unique(idx[:, 2]).numel())}; GEMM right: {gem_ok}\n      e.g. sample {s0} row {r0} head {h0}: solo {solo.view(B, L, VH, 3)[s0, r0, h0].tolist()} beside {out.view(B, L, VH, 3)[s0, r0, h0].tolist()}", flush=True)
