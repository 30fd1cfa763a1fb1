"""r06: the reduced victim (scratch/ubench/pk_rotate.hip: loads -> 3x3 rotations -> store) alone and beside the 256x256 GEMM on another
stream.  Per configuration: rounds whose output differs from the solo run, waves hit, by lane quarter, by 64-row trip, by repetition."""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
B, L, VH = int(os.environ.get("NB_B", "50")), int(os.environ.get("NB_L", "258")), 256
ROUNDS = int(os.environ.get("ROUNDS", "12"))
SLEEP = int(os.environ.get("SLEEP_CYCLES", "1500000"))
gd = torch.Generator(device="cuda").manual_seed(0)
P = (torch.randn(B * L, 15 * VH, generator=gd, device="cuda") * 0.7).contiguous()
rot = torch.linalg.qr(torch.randn(B * L, 3, 3, generator=gd, device="cuda"))[0].reshape(B * L, 9).contiguous()
trans = (torch.randn(B * L, 3, generator=gd, device="cuda") * 10).contiguous()
fmask = torch.ones(B * L, dtype=torch.uint8, device="cuda")
M = 12900
A = (torch.rand(M, 1536, generator=gd, device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(3840, 1536, generator=gd, device="cuda") * 2 - 1) / 39.0).to(torch.bfloat16)
og = gemm_bf16(A, W, N.EPI_BF16).clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
stride = B * L * 3 * VH
for cfg in os.environ.get("CONFIGS", "pk:64:1:12480,nopk:64:1:12480,pk:256:1:12480,pk:64:8:12480,pk:64:1:0,pk:64:1:40960").split(","):
    tag, nt, reps, lds = cfg.split(":")
    nt, reps, lds = int(nt), int(reps), int(lds)
    if tag.startswith("co="):          # an ISA-level variant of rotate_kernel<64> (scratch/ubench/isa/mk_rotate.py), through the module API
        hip = ctypes.CDLL("libamdhip64.so")
        mod, kfn = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), os.path.join(ROOT, f"scratch/ubench/isa/rot_{tag[3:]}.co").encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(kfn), mod, b"_Z13rotate_kernelILi64EEvPKfS1_S1_PKhPfiiil") == 0
        assert nt == 64
        def run(out, stream):
            vals = [ctypes.c_void_p(t.data_ptr()) for t in (P, rot, trans, fmask, out)] + [ctypes.c_int(L), ctypes.c_int(VH), ctypes.c_int(reps), ctypes.c_int64(stride)]
            params = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.pointer(v), ctypes.c_void_p) for v in vals])
            rc = hip.hipModuleLaunchKernel(kfn, VH, B, 1, 64, 1, 1, lds, ctypes.c_void_p(stream), params, None)
            assert rc == 0, rc
            return vals
    else:
        lib = ctypes.CDLL(os.path.join(ROOT, "scratch/ubench", "pk_rotate.so" if tag == "pk" else f"pk_rotate_{tag}.so"))
        fn = getattr(lib, f"pr_rotate_{tag}")
        fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        def run(out, stream):
            rc = fn(P.data_ptr(), rot.data_ptr(), trans.data_ptr(), fmask.data_ptr(), out.data_ptr(), B, L, VH, reps, stride, nt, lds, stream)
            assert rc == 0, rc
    solo = torch.zeros(reps, B, L, VH, 3, device="cuda")
    run(solo, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(solo, torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    hits, waves, quarters, trips, byrep = 0, 0, [0] * 4, [0] * 5, [0] * reps
    for it in range(ROUNDS):
        out = torch.zeros_like(solo)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(4):
                gemm_bf16(A, W, N.EPI_BF16, out=og)
        with torch.cuda.stream(s2):
            torch.cuda._sleep(SLEEP)
            keep = run(out, s2.cuda_stream)
        with torch.cuda.stream(s1):
            for _ in range(12):
                gemm_bf16(A, W, N.EPI_BF16, out=og)
        torch.cuda.synchronize()
        d = (out.view(torch.int32) != solo.view(torch.int32)).any(-1)           # (rep, sample, row, head)
        if bool(d.any()):
            idx = torch.nonzero(d)
            hits += 1
            waves += int(torch.unique((idx[:, 1] * VH + idx[:, 3]) * 8 + idx[:, 2] // 64 % 4 * (nt == 256)).numel())
            quarters = [a + b for a, b in zip(quarters, torch.bincount((idx[:, 2] % 64) // 16, minlength=4).tolist())]
            trips = [a + b for a, b in zip(trips, torch.bincount(idx[:, 2] // 64, minlength=5).tolist())]
            byrep = [a + b for a, b in zip(byrep, torch.bincount(idx[:, 0], minlength=reps).tolist())]
            if os.environ.get("DUMP") and hits <= 2:
                # what did the wrong lanes compute?  a0 = R0 v0 + R1 v1 + R2 v2, a1 = R3 v0 + R4 v1 + R5 v2, a2 = R6 v0 + R7 v1 + R8 v2 (float64 here)
                for r_, b_, q_, h_ in idx[:4].tolist():
                    row = b_ * L + q_
                    R = rot[row].double(); v = P[row, 3 * h_:3 * h_ + 3].double()
                    prev = P[row - 64, 3 * h_:3 * h_ + 3].double() if q_ >= 64 else torch.zeros(3, dtype=torch.float64, device="cuda")
                    good, bad = solo[r_, b_, q_, h_].double(), out[r_, b_, q_, h_].double()
                    hyp = {
                        "right": torch.stack([R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2], R[6] * v[0] + R[7] * v[1] + R[8] * v[2]]),
                        "#6 without its op_sel (lo: R1 v0, hi: R3 v1)": torch.stack([R[0] * v[0] + R[1] * v[0] + R[2] * v[2], R[3] * v[1] + R[4] * v[1] + R[5] * v[2], R[6] * v[0] + R[7] * v[1] + R[8] * v[2]]),
                        "#6 dropped (no R1 v1 / R3 v0 term)": torch.stack([R[0] * v[0] + R[2] * v[2], R[4] * v[1] + R[5] * v[2], R[6] * v[0] + R[7] * v[1] + R[8] * v[2]]),
                        "#6 with v of the row 64 before": torch.stack([R[0] * v[0] + R[1] * prev[1] + R[2] * v[2], R[3] * prev[0] + R[4] * v[1] + R[5] * v[2], R[6] * v[0] + R[7] * v[1] + R[8] * v[2]]),
                        "#6 hi half = lo half's product (R1 v1 twice)": torch.stack([R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[1] * v[1] + R[4] * v[1] + R[5] * v[2], R[6] * v[0] + R[7] * v[1] + R[8] * v[2]]),
                    }
                    print(f"   sample {b_} row {q_} head {h_}: alone {[round(x, 6) for x in good.tolist()]} beside {[round(x, 6) for x in bad.tolist()]}")
                    for k_, val in hyp.items():
                        print(f"      {k_:48s} {[round(x, 6) for x in val.tolist()]}  max |diff to beside| {float((val - bad).abs().max()):.2e}")
    print(f"[{tag}, {nt} threads per workgroup, {reps} repetition(s), LDS request {lds} B] alone {e0.elapsed_time(e1) * 1e3:.0f} us; rounds with differing outputs {hits} of {ROUNDS}; "
          f"waves hit {waves}; outputs by lane quarter {quarters}; by 64-row trip {trips}; by repetition {byrep}", flush=True)
