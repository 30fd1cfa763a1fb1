"""r06: ISA-level variants of geom_attention_kernel<float> (scratch/ubench/isa/mk.py: the compiler's assembly with edits, as code
objects) beside the 256x256 GEMM on another stream; counts the (sample, row, head) outputs that differ from the variant's solo run."""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
from esmdiff_amd.geometry import build_affine3d_from_coordinates
hip = ctypes.CDLL("libamdhip64.so")
B, L, VH = int(os.environ.get("NB_B", "50")), int(os.environ.get("NB_L", "258")), 256
ROUNDS = int(os.environ.get("ROUNDS", "12"))
SLEEP = int(os.environ.get("SLEEP_CYCLES", "1500000"))
LDS = int(os.environ.get("NB_LDS", str(((L + 3) & ~3) * 48)))
NEIGHBOUR = os.environ.get("NEIGHBOUR", "gemm")
NB_ITERS = int(os.environ.get("NB_ITERS", "600"))
nbl = ctypes.CDLL(os.path.join(ROOT, "scratch/ubench/mfma_neighbour.so"))
nbl.nb_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
KNAME = b"_ZN2ed21geom_attention_kernelIfEEvPKT_PKfS5_PKhS5_S5_PS1_ii"
g = torch.Generator().manual_seed(1)
ca = torch.cumsum(torch.nn.functional.normalize(torch.randn(L, 3, generator=g), dim=-1) * 3.8, 0)
xyz = torch.stack([ca + torch.tensor([-1.2, 0.7, 0.0]), ca, ca + torch.tensor([1.3, 0.6, 0.1])], 1)
for part in [p for p in os.environ.get("FRAMELESS", "0,97:161,257").split(",") if p]:      # rows without a frame
    lo, _, hi = part.partition(":")
    xyz[int(lo):int(hi or int(lo) + 1)] = float("nan")
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
sink = torch.zeros(1024, device="cuda")
for tag in os.environ.get("VARIANTS", "i0,i1,i2,i3,i4").split(","):
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    rc = hip.hipModuleLoad(ctypes.byref(mod), os.path.join(ROOT, f"scratch/ubench/isa/geom_{tag}.co").encode())
    assert rc == 0, ("hipModuleLoad", rc)
    rc = hip.hipModuleGetFunction(ctypes.byref(fn), mod, KNAME)
    assert rc == 0, ("hipModuleGetFunction", rc)
    def run(out, stream):
        vals = [ctypes.c_void_p(t.data_ptr()) for t in (P, rot, trans, fmask, w_rot, w_dist, out)] + [ctypes.c_int(L), ctypes.c_int(VH)]
        params = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.pointer(v), ctypes.c_void_p) for v in vals])
        rc = hip.hipModuleLaunchKernel(fn, VH, B, 1, 64, 1, 1, LDS, ctypes.c_void_p(stream), params, None)
        assert rc == 0, ("launch", rc)
        return vals
    solo = torch.zeros(B * L, 3 * VH, device="cuda")
    keep = run(solo, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        keep = run(solo, torch.cuda.current_stream().cuda_stream)
    e1.record(); torch.cuda.synchronize()
    print(f"[{tag}] alone: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per launch (B {B}, L {L}, LDS request {LDS} B)", flush=True)
    hits, wgs, quarters_all, trips_all = 0, 0, [0] * 4, [0] * 8
    for it in range(ROUNDS):
        out = torch.zeros_like(solo)
        torch.cuda.synchronize()
        if NEIGHBOUR == "gemm":
            with torch.cuda.stream(s1):
                for _ in range(4):
                    gemm_bf16(A, W, N.EPI_BF16, out=og)
            with torch.cuda.stream(s2):
                torch.cuda._sleep(SLEEP)
                keep = run(out, s2.cuda_stream)
            with torch.cuda.stream(s1):
                for _ in range(12):
                    gemm_bf16(A, W, N.EPI_BF16, out=og)
        else:
            kind, span = NEIGHBOUR.split("_")          # mfma|hold _ long|short
            mode = 1 if kind == "mfma" else 0
            launches, iters = (1, NB_ITERS * 60) if span == "long" else (60, NB_ITERS)
            with torch.cuda.stream(s2):
                torch.cuda._sleep(SLEEP)
                keep = run(out, s2.cuda_stream)
            with torch.cuda.stream(s1):
                for _ in range(launches):
                    assert nbl.nb_launch(mode, iters, 128 * 1024, 256, sink.data_ptr(), s1.cuda_stream) == 0
        torch.cuda.synchronize()
        d = (out.view(torch.int32) != solo.view(torch.int32)).view(B, L, VH, 3).any(-1)
        if bool(d.any()):
            idx = torch.nonzero(d)
            hits += 1
            wgs += int(torch.unique(idx[:, 0] * VH + idx[:, 2]).numel())
            q = torch.bincount((idx[:, 1] % 64) // 16, minlength=4).tolist()
            t = torch.bincount(idx[:, 1] // 64, minlength=8).tolist()
            quarters_all = [a + b for a, b in zip(quarters_all, q)]
            trips_all = [a + b for a, b in zip(trips_all, t)]
    print(f"[{tag}] L {L} frameless {os.environ.get('FRAMELESS', 'default')} neighbour {NEIGHBOUR}: rounds with differing outputs {hits} of {ROUNDS}; workgroups hit {wgs}; by lane quarter {quarters_all}; by query trip {trips_all}; solo finite {bool(torch.isfinite(solo).all())}", flush=True)
