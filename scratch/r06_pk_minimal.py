"""r06: the minimal victim (scratch/ubench/pk_minimal.hip: global load -> full wait -> one packed and two plain float multiplies,
compared in the kernel) alone and beside the 256x256 GEMM on another stream."""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
lib = ctypes.CDLL(os.path.join(ROOT, "scratch/ubench/pk_minimal.so"))
lib.pkmin_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
B, VH = 50, 256
REPS = int(os.environ.get("REPS", "600"))
LDS = int(os.environ.get("NB_LDS", "12480"))
ROUNDS = int(os.environ.get("ROUNDS", "8"))
gd = torch.Generator(device="cuda").manual_seed(0)
G = torch.randn(1 << 24, generator=gd, device="cuda")
M = 12900
A = (torch.rand(M, 1536, generator=gd, device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(3840, 1536, generator=gd, device="cuda") * 2 - 1) / 39.0).to(torch.bfloat16)
og = gemm_bf16(A, W, N.EPI_BF16).clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def launch(err, stream):
    assert lib.pkmin_launch(G.data_ptr(), G.numel() // 4, B, VH, REPS, LDS, err.data_ptr(), stream) == 0
err = torch.zeros(16, dtype=torch.int32, device="cuda")
launch(err, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
print("alone:", err.tolist(), flush=True)
for it in range(ROUNDS):
    err.zero_(); torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        for _ in range(4):
            gemm_bf16(A, W, N.EPI_BF16, out=og)
    with torch.cuda.stream(s2):
        torch.cuda._sleep(int(os.environ.get("SLEEP_CYCLES", "1500000")))
        launch(err, s2.cuda_stream)
    with torch.cuda.stream(s1):
        for _ in range(12):
            gemm_bf16(A, W, N.EPI_BF16, out=og)
    torch.cuda.synchronize()
    e = err.tolist()
    print(f"round {it}: mismatching (lane, rep) by lane quarter {e[:4]}; lanes with a mismatch {e[12]}; first mismatch of a lane by eighth of the run {e[4:12]}", flush=True)
