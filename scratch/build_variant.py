"""Builds A/B variants of libesmdiff_hip.so that differ in the -D flags of ONE translation unit (run here, on the CPU
container; the .so files travel to the GPU box with the snapshot):

    python scratch/build_variant.py gemm256 abl8 -DED_ABL=8      ->  esmdiff_amd/lib/libesmdiff_hip_abl8.so
    ESMDIFF_LIB=esmdiff_amd/lib/libesmdiff_hip_abl8.so python scratch/bench_gemm.py
"""
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from esmdiff_amd import build as B   # noqa: E402

unit, tag, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
obj = B.OBJDIR / f"{unit}_{tag}.o"
cmd = [B._hipcc(), *B.COMMON, *B.UNITS[unit], *flags, "-c", str(B.CSRC / f"{unit}.hip"), "-o", str(obj)]
subprocess.run(cmd, check=True)
objs = [str(obj if n == unit else B.OBJDIR / f"{n}.o") for n in B.UNITS] + [str(B.OBJDIR / f"{n}_f16.o") for n in B.F16_UNITS]
out = B.LIBDIR / f"libesmdiff_hip_{tag}.so"
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(out)], check=True)
print(out)
