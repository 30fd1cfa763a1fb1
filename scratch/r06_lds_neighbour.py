"""r06: the victim of "one-wave LDS kernel next to a 128 KB LDS-DMA GEMM workgroup on the same CU" (profiles/r06_frames_two_queue_race.txt).
Stream 1 runs the real 256x256 GEMM (block 0's geometric proj shape) several times, stream 2 a checker kernel of
geom_attention_kernel's shape that verifies every global and LDS word it reads (scratch/ubench/lds_neighbour.hip).  Per neighbour
variant: wrong GEMM rows against a solo run, wrong global / LDS words seen by the checker."""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
nb = ctypes.CDLL(os.path.join(ROOT, "scratch/ubench/lds_neighbour.so"))
nb.nb_fill.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
nb.nb_checker.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]
nb.nb_nolds.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]

B, L, VH = 50, 258, 256
M, ldp = B * L, 15 * VH
ROUNDS = int(os.environ.get("ROUNDS", "12"))
NG = int(os.environ.get("NG", "6"))
g = torch.Generator(device="cuda").manual_seed(0)
G = torch.empty(M * ldp, dtype=torch.int16, device="cuda")
assert nb.nb_fill(G.data_ptr(), G.numel(), torch.cuda.current_stream().cuda_stream) == 0
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
shapes = (("geom proj", 3840, 1536), ("geom out", 1536, 768), ("qkv", 4608, 1536))
variants = (("nolds", None), ("checker 12.5 KB", 0), ("checker 16 KB", 16384), ("checker 40 KB", 40960))
for name, Nn, K in shapes:
    A = (torch.rand(M, K, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(Nn, K, generator=g, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    ref = gemm_bf16(A, W, N.EPI_BF16).clone()
    outs = [torch.empty_like(ref) for _ in range(NG)]
    torch.cuda.synchronize()
    for vname, lds in variants:
        err = torch.zeros(16, dtype=torch.int32, device="cuda")
        wrong, worst, t_g, t_c = [], 0.0, 0.0, 0.0
        for it in range(ROUNDS):
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            with torch.cuda.stream(s2):
                e2.record()
                for _ in range(2):
                    rc = nb.nb_nolds(B, L, VH, 1, err.data_ptr(), s2.cuda_stream) if lds is None else \
                        nb.nb_checker(G.data_ptr(), ldp, B, L, VH, lds, 1, err.data_ptr(), s2.cuda_stream)
                    assert rc == 0, rc
                e3.record()
            with torch.cuda.stream(s1):
                e0.record()
                for o in outs:
                    gemm_bf16(A, W, N.EPI_BF16, out=o)
                e1.record()
            torch.cuda.synchronize()
            t_g += e0.elapsed_time(e1); t_c += e2.elapsed_time(e3)
            for gi, o in enumerate(outs):
                d = (o != ref).any(1)
                if bool(d.any()):
                    rows = torch.nonzero(d).flatten()
                    worst = max(worst, float((o.float() - ref.float()).abs().max()))
                    wrong.append((it, gi, int(d.sum()), int(rows.min()), int(rows.max())))
        e = err.cpu().tolist()
        print(f"{name:10s} N={Nn} K={K} | neighbour {vname:16s} | GEMM launches with wrong rows {len(wrong)} of {ROUNDS * NG} (largest |diff| {worst:.3g}; first {wrong[:3]}) | "
              f"checker: wrong global words {e[0]}, wrong LDS words {e[1]}" + (f", first (wg, key, word, got, want, rep) = {[hex(v & 0xffffffff) for v in e[2:8]]}" if e[1] else "")
              + f" | per round: {NG} GEMMs {t_g / ROUNDS:.2f} ms, 2 neighbours {t_c / ROUNDS:.2f} ms", flush=True)
