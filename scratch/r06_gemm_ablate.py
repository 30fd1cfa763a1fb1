"""r06: A/B libraries whose 256x256 GEMM lacks parts beyond -DED_ABL4's bits, WITHOUT touching csrc/gemm256w4.hip (its hash is part of
the traffic record): a patched copy is compiled to an object that replaces gemm256w4.o in the link.
    python scratch/r06_gemm_ablate.py <tag> <ED_ABL4 bits> [nobarrier] [novmwait] [nolgkmwait] [nozeroform]
      -> esmdiff_amd/lib/libesmdiff_hip_<tag>.so        (wrong GEMM results by construction; neighbours for scratch/r06_pk_opsel.py)"""
import subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd import build as B
tag, bits, opts = sys.argv[1], sys.argv[2], set(sys.argv[3:])
B.build()
src = (B.CSRC / "gemm256w4.hip").read_text()
def sub(old, new, count=None):
    global src
    assert old in src, old
    src = src.replace(old, new) if count is None else src.replace(old, new, count)
if "nobarrier" in opts:
    sub("    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // this wave's share of the next K-tile has landed\n    __builtin_amdgcn_s_barrier();",
        "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");")
if "novmwait" in opts:
    sub("    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // this wave's share of the next K-tile has landed\n", "")
if "nolgkmwait" in opts:
    sub("  asm volatile(\"s_waitcnt lgkmcnt(0)\"                                                                              \\", "  asm volatile(\"\"                                                                              \\")
if "nozeroform" in opts:
    sub("    if constexpr (decltype(FIRST)::value) W4_MFMA0(acc[i_][j_], C.b[j_], C.a[i_]);               \\\n    else W4_MFMA(acc[i_][j_], C.b[j_], C.a[i_]);", "    W4_MFMA(acc[i_][j_], C.b[j_], C.a[i_]);")
tmp = B.CSRC / f"_ablate_{tag}_gemm256w4.hip"
tmp.write_text(src)
obj = B.OBJDIR / f"gemm256w4_{tag}.o"
try:
    subprocess.run([B._hipcc(), *B.COMMON, *B.UNITS["gemm256w4"], f"-DED_ABL4={bits}", "-c", str(tmp), "-o", str(obj)], check=True)
finally:
    tmp.unlink()
objs = [str(obj if n == "gemm256w4" else B.OBJDIR / f"{n}.o") for n in B.UNITS] + [str(B.OBJDIR / f"{n}_f16.o") for n in B.F16_UNITS]
out = B.LIBDIR / f"libesmdiff_hip_{tag}.so"
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(out)], check=True)
print(out)
