"""r06: geom_attention_kernel's key walk alone (scratch/ubench/pk_neighbour.hip), compiled with and without packed float ops, run
alone and then beside the 256x256 GEMM on another stream: which (workgroup, lane) outputs differ from the solo run, in which of
the five per-lane values, and by how much."""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
B, L, VH = int(os.environ.get("NB_B", "50")), 258, 256
M = 12900
ROUNDS = int(os.environ.get("ROUNDS", "6"))
LDS = int(os.environ.get("NB_LDS", "0"))
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.rand(M, 1536, generator=g, device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(3840, 1536, generator=g, device="cuda") * 2 - 1) / 39.0).to(torch.bfloat16)
ref_g = gemm_bf16(A, W, N.EPI_BF16).clone()
og = torch.empty_like(ref_g)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for tag in os.environ.get("LIBS", "pk,nopk").split(","):
    lib = ctypes.CDLL(os.path.join(ROOT, "scratch/ubench", "pk_neighbour.so" if tag == "pk" else f"pk_neighbour_{tag}.so"))
    fn = getattr(lib, f"nb_walk_{tag}")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    solo = torch.zeros(B * VH * L, 8, device="cuda")
    assert fn(solo.data_ptr(), B, L, VH, LDS, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    again = torch.zeros_like(solo)
    assert fn(again.data_ptr(), B, L, VH, LDS, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    print(f"[{tag}] solo twice equal: {bool(torch.equal(solo.view(torch.int32), again.view(torch.int32)))}", flush=True)
    for it in range(ROUNDS):
        out = torch.zeros_like(solo)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(4):
                gemm_bf16(A, W, N.EPI_BF16, out=og)
        with torch.cuda.stream(s2):
            torch.cuda._sleep(int(os.environ.get("SLEEP_CYCLES", "1500000")))
            assert fn(out.data_ptr(), B, L, VH, LDS, s2.cuda_stream) == 0
        with torch.cuda.stream(s1):
            for _ in range(12):
                gemm_bf16(A, W, N.EPI_BF16, out=og)
        torch.cuda.synchronize()
        d = out.view(torch.int32) != solo.view(torch.int32)           # (wg*L+q, 8)
        bad = torch.nonzero(d.any(1)).flatten()
        gem_ok = bool(torch.equal(og, ref_g))
        if bad.numel() == 0:
            print(f"[{tag}] round {it}: equal to the solo run (GEMM right: {gem_ok})", flush=True)
            continue
        q = bad % L
        wg = bad // L
        lanes = torch.bincount(q % 64, minlength=64)
        quarters = [int(lanes[i * 16:(i + 1) * 16].sum()) for i in range(4)]
        its = torch.bincount(q // 64, minlength=5).tolist()
        cols = d[bad].sum(0).tolist()[:5]
        i0 = int(bad[0])
        print(f"[{tag}] round {it}: {bad.numel()} (workgroup, query) outputs differ in {int(torch.unique(wg).numel())} workgroups (first {int(wg.min())}, last {int(wg.max())}); "
              f"by lane quarter {quarters}; by query trip {its}; by value [m, den, o0, o1, o2] {cols}; GEMM right: {gem_ok}\n"
              f"      e.g. wg {i0 // L} q {i0 % L}: solo {solo[i0, :5].tolist()} beside {out[i0, :5].tolist()}", flush=True)
