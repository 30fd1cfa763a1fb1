"""r06: the self-checking minimal victim (scratch/ubench/pk_opsel.hip) alone and beside the 256x256 GEMM on another stream, per packed form."""
import ctypes, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esmdiff_amd import _native as N
from esmdiff_amd.engine import gemm_bf16
lib = ctypes.CDLL(os.path.join(ROOT, "scratch/ubench/pk_opsel.so"))
lib.opsel_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
# few, long-lived victim waves (2 one-wave workgroups per CU) that stay beside the neighbour's workgroups as those come and go
B, VH = int(os.environ.get("NB_B", "2")), 256
REPS = int(os.environ.get("REPS", "3000"))
NAGG = int(os.environ.get("NAGG", "24"))
LDS = int(os.environ.get("NB_LDS", "12480"))
ROUNDS = int(os.environ.get("ROUNDS", "10"))
gd = torch.Generator(device="cuda").manual_seed(0)
G = torch.randn(1 << 24, generator=gd, device="cuda")
M = 12900
A = (torch.rand(M, 1536, generator=gd, device="cuda") * 2 - 1).to(torch.bfloat16)
W = ((torch.rand(3840, 1536, generator=gd, device="cuda") * 2 - 1) / 39.0).to(torch.bfloat16)
og = gemm_bf16(A, W, N.EPI_BF16).clone()
AGGRESSOR = os.environ.get("AGGRESSOR", "ours")      # ours: the product's 256x256 GEMM; torch: torch.matmul (hipBLASLt / rocBLAS) on the same operands
Wt = W.t().contiguous()
nbl = ctypes.CDLL(os.path.join(ROOT, "scratch/ubench", os.environ.get("NB_LIB", "mfma_neighbour.so")))
nbl.nb_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(1024, device="cuda")
def aggress():
    if AGGRESSOR == "ours":
        gemm_bf16(A, W, N.EPI_BF16, out=og)
    elif AGGRESSOR == "torch":
        torch.matmul(A, Wt, out=og)
    else:     # synth1 MFMA only, synth2 MFMA + ds_read_b128 (the GEMM's k-step), synth3 ds_read_b128 only, synth0 registers held: scratch/ubench/mfma_neighbour.hip
        assert nbl.nb_launch(int(AGGRESSOR[5:]), 1200, 128 * 1024, 256, sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
FORMS = ["fma op_sel:[0,1,0] op_sel_hi:[1,0,1]", "fma plain", "fma op_sel_hi:[1,0,1]", "mul op_sel:[0,1] op_sel_hi:[1,0]", "add A,A op_sel:[0,1] op_sel_hi:[1,0]", "fma op_sel:[1,0,0]", "fma op_sel:[0,0,1]", "fma op_sel:[0,1,0]", "mul s[..],B op_sel:[1,0]", "fma_f16 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "mul_f16 op_sel:[0,1] op_sel_hi:[1,0]", "add_f16 op_sel:[0,1] op_sel_hi:[1,0]"]
for mode in [int(m) for m in os.environ.get("MODES", "0,1,2,3,4,5,6,7,8").split(",")]:
    def launch(err, stream):
        assert lib.opsel_launch(mode, G.data_ptr(), G.numel() // 8, B, VH, REPS, LDS, err.data_ptr(), stream) == 0
    err = torch.zeros(16, dtype=torch.int32, device="cuda")
    launch(err, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    alone = err.tolist()[:8]
    tot = [0] * 8
    hit_rounds = 0
    for it in range(ROUNDS):
        err.zero_(); torch.cuda.synchronize()
        ev0, ev1, ev2, ev3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        with torch.cuda.stream(s2):
            ev2.record(); launch(err, s2.cuda_stream); ev3.record()
        with torch.cuda.stream(s1):
            ev0.record()
            for _ in range(NAGG):
                aggress()
            ev1.record()
        torch.cuda.synchronize()
        e = err.tolist()[:8]
        t_v, t_a = ev2.elapsed_time(ev3), ev0.elapsed_time(ev1)
        hit_rounds += int(sum(e[:4]) > 0)
        tot = [a + b for a, b in zip(tot, e)]
    print(f"v_pk_{FORMS[mode]:40s} alone: {sum(alone[:4])} mismatches; beside the GEMM ({AGGRESSOR}): rounds with mismatches {hit_rounds} of {ROUNDS}; (lane, rep) mismatches by lane quarter {tot[:4]}; "
          f"lo half {tot[4]}, hi half {tot[5]}; lo half == C.lo (mul: == +-0) exactly {tot[6]}; lanes {tot[7]}; last round: victim {t_v:.2f} ms, {NAGG} neighbour launches {t_a:.2f} ms", flush=True)
