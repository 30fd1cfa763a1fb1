"""Builds esmdiff_amd/lib/libesmdiff_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m esmdiff_amd.build [--force]

One translation unit per kernel family; sampler.hip is compiled with -ffp-contract=off so that
csrc/ed_math.h evaluates bit-identically on host and device (see oracle/csrc/sampler_oracle.c).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
OBJDIR = PKG / "build"
LIB = LIBDIR / "libesmdiff_hip.so"
INCLUDE = PKG.parent / "include"

UNITS = {
    "engine": [],
    "gemm": [],
    "gemm256w4": [],
    "gemm_split": [],
    "attention_split": [],
    "strict": [],
    # no packed float ops in the geometric kernel: v_pk_fma_f32 / v_pk_mul_f32 with op_sel[1] = 1 (the SLP form of its 3x3 rotations)
    # compute lanes 48-63 wrong next to the other queue's GEMM (csrc/geom.hip, profiles/r06_frames_two_queue_race.txt);
    # tests/test_host_cpu.py scans every code object of the shipped library for that form
    "geom": ["-fno-slp-vectorize", "-fno-vectorize"],
    "encoder": [],
    "attention": [],
    "norm": [],
    "embed": [],
    "pairwise": [],
    "convert": [],
    "sampler": ["-ffp-contract=off"],
    "gibbs": ["-ffp-contract=off"],
    "metrics": ["-ffp-contract=off"],
}
# The MFMA kernels, the row kernels that feed them and the weight conversion are compiled a SECOND time with f16 operands
# (csrc/ed_half.h): -DED_F16 switches the conversion / MFMA primitives, -Ded=ed16 puts that build into namespace ed16
# (every `ed` token of those sources is the namespace name).  precision="f16" engines call the ed16 kernels.
F16_UNITS = ["gemm", "gemm256w4", "attention", "norm", "geom", "convert"]
F16_FLAGS = ["-DED_F16", "-Ded=ed16"]
EXTRA = os.environ.get("ESMDIFF_EXTRA_CXXFLAGS", "").split()
COMMON = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


HEADERS = [INCLUDE / "esmdiff_hip.h", INCLUDE / "esmdiff_hip_test.h"]


def _newest_src() -> float:
    files = list(CSRC.glob("*")) + HEADERS + [Path(__file__)]
    return max(f.stat().st_mtime for f in files)


def _write_export_map(path: Path) -> None:
    """The dynamic symbol table holds the C ABI and nothing else: every `esmdiff_*` function the two headers declare is global,
    everything else (the C++ launchers of both operand-type namespaces, template instantiations, the HIP registration
    helpers) is local to the library.  tests/test_host_cpu.py checks `nm -D`."""
    import re
    names = set()
    for h in HEADERS:
        names |= set(re.findall(r"\b(esmdiff_[a-z0-9_]+)\s*\(", h.read_text()))
    names -= {"esmdiff_gemm_epilogue"}
    path.write_text("{\n  global:\n" + "".join(f"    {n};\n" for n in sorted(names)) + "  local:\n    *;\n};\n")


def build(force: bool = False, verbose: bool = False) -> Path:
    LIBDIR.mkdir(exist_ok=True)
    OBJDIR.mkdir(exist_ok=True)
    stamp = OBJDIR / "flags.txt"
    flags = " ".join(COMMON) + " | " + " ".join(f"{n}: {' '.join(f)}" for n, f in UNITS.items() if f)
    if not stamp.exists() or stamp.read_text() != flags:   # different flags (e.g. -DED_GEMM_DEBUG, a unit's own flags): rebuild all
        force = True
    if LIB.exists() and not force and LIB.stat().st_mtime >= _newest_src():
        return LIB
    cc = _hipcc()
    headers = [p.stat().st_mtime for p in list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + HEADERS]

    def compile_one(job):
        name, f16 = job
        src, obj = CSRC / f"{name}.hip", OBJDIR / (f"{name}_f16.o" if f16 else f"{name}.o")
        if (not force and obj.exists() and obj.stat().st_mtime >= max([src.stat().st_mtime] + headers)):
            return name, ""
        cmd = [cc, *COMMON, *UNITS[name], *(F16_FLAGS if f16 else []), "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
        return name, r.stderr

    jobs = [(n, False) for n in UNITS] + [(n, True) for n in F16_UNITS]
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        for name, err in ex.map(compile_one, jobs):
            if verbose and err.strip():
                print(f"[{name}] {err}", file=sys.stderr)
    objs = [str(OBJDIR / f"{n}.o") for n in UNITS] + [str(OBJDIR / f"{n}_f16.o") for n in F16_UNITS]
    emap = OBJDIR / "exports.map"
    _write_export_map(emap)
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={emap}", *objs, "-o", str(LIB)],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    stamp.write_text(flags)
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print(p)
