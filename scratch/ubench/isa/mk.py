"""r06: ISA-level variants of geom_attention_kernel for scratch/r06_geom_isa.py (run here, on the CPU container; the code objects
travel to the GPU box).  Base = hipcc's own assembly of csrc/geom.hip WITH SLP vectorisation (the packed float ops the product
build now switches off); variants = textual edits of it, assembled with clang and linked with ld.lld into geom_<tag>.co.

  i0    unmodified                                   nopk  compiled with -fno-slp-vectorize -fno-vectorize (the product build)
  i1    every s_waitcnt is a full wait               i2    8 idle cycles behind every packed float op
  i3/4  every lgkmcnt / vmcnt wait is a full one     i5    16 idle cycles behind every lgkmcnt wait
  i7    8 idle cycles in front of every packed op    i8    8 idle cycles behind every ds_read
  i9    128 idle cycles behind every vmcnt wait      i10   512 idle cycles behind every vmcnt wait
  d1..5 source-level probes: the kernel stores q_rot (d1), q_dist (d2), (m, den, o0) (d3), sums of the loaded rotation and
        translation (d4) or the unnormalised accumulators (d5) instead of its result (the walk stays alive through 0 * o terms)"""
import re, subprocess, sys
from pathlib import Path
HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[2]
LLVM = "/opt/rocm/lib/llvm/bin/"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         f"-I{ROOT / 'include'}", f"-I{ROOT / 'esmdiff_amd/csrc'}", "-S", "--cuda-device-only"]
SRC = ROOT / "esmdiff_amd/csrc/geom.hip"


def asm_of(src, extra=()):
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, str(src), "-o", "/dev/stdout"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def build(tag, text):
    (HERE / f"geom_{tag}.s").write_text(text)
    subprocess.run([LLVM + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(HERE / f"geom_{tag}.s"), "-o", str(HERE / f"geom_{tag}.o")], check=True)
    subprocess.run([LLVM + "ld.lld", "-shared", str(HERE / f"geom_{tag}.o"), "-o", str(HERE / f"geom_{tag}.co")], check=True)
    print(tag, "built;", len(re.findall(r"\bv_pk_\w+", text)), "packed ops")


base = asm_of(SRC)
build("i0", base)
build("nopk", asm_of(SRC, ["-fno-slp-vectorize", "-fno-vectorize"]))
build("i1", re.sub(r"s_waitcnt [a-z]+cnt\(\d+\)( [a-z]+cnt\(\d+\))*", "s_waitcnt vmcnt(0) lgkmcnt(0)", base))
build("i2", re.sub(r"(\n\tv_pk_[a-z0-9_]+ [^\n]*)", r"\1\n\ts_nop 7", base))
build("i3", re.sub(r"s_waitcnt lgkmcnt\(\d+\)", "s_waitcnt lgkmcnt(0)", base))
build("i4", re.sub(r"s_waitcnt vmcnt\(\d+\)", "s_waitcnt vmcnt(0)", base))
build("i5", re.sub(r"(\n\ts_waitcnt lgkmcnt\(\d+\))", r"\1\n\ts_nop 7\n\ts_nop 7", base))
build("i7", re.sub(r"(\n\tv_pk_[a-z0-9_]+ )", r"\n\ts_nop 7\1", base))
build("i8", re.sub(r"(\n\tds_read[a-z0-9_]* [^\n]*)", r"\1\n\ts_nop 7", base))
build("i9", re.sub(r"(\n\ts_waitcnt vmcnt\(\d+\))", r"\1" + "\n\ts_nop 7" * 16, base))
build("i10", re.sub(r"(\n\ts_waitcnt vmcnt\(\d+\))", r"\1" + "\n\ts_nop 7" * 64, base))
text = SRC.read_text()
old = "    float r0 = 0.f, r1 = 0.f, r2 = 0.f;\n    if (den > 0.f) {"
assert old in text
probes = {
    "d1": "r0 = qr[0] + 0.f * o0; r1 = qr[1] + 0.f * o1; r2 = qr[2] + 0.f * o2 + 0.f * den + 0.f * m;",
    "d2": "r0 = qd[0] + 0.f * o0; r1 = qd[1] + 0.f * o1; r2 = qd[2] + 0.f * o2 + 0.f * den + 0.f * m;",
    "d3": "r0 = m + 0.f * o1; r1 = den + 0.f * o2; r2 = o0;",
    "d4": "r0 = R[0] + R[3] + R[6] + 0.f * o0; r1 = R[1] + R[4] + R[7] + 0.f * o1; r2 = R[2] + R[5] + R[8] + t[0] + t[1] + t[2] + 0.f * o2 + 0.f * den + 0.f * m;",
    "d5": "r0 = o0; r1 = o1; r2 = o2 + 0.f * den + 0.f * m;",
}
tmp = HERE / "_probe.hip"
for tag, code in probes.items():
    tmp.write_text(text.replace(old, "    float r0 = 0.f, r1 = 0.f, r2 = 0.f;\n    if (den > 0.f) { " + code + " den = 0.f; }\n    if (den > 0.f) {"))
    build(tag, asm_of(tmp))
tmp.unlink()
# i11 / i12: the float32 kernel's register allocation padded to 104 (still fits beside a 408-register GEMM wave on a SIMD) / 128 (does not)
for tag, n in (("i11", 104), ("i12", 128)):
    t, k = re.subn(r"(\.amdhsa_next_free_vgpr) 60(\n[^\n]*\n\s*\.amdhsa_accum_offset) 60", rf"\1 {n}\2 {n}", base)
    assert k == 1
    build(tag, t.replace(".vgpr_count:     60", f".vgpr_count:     {n}"))
# e1: d1 (stores q_rot) without the key walk; e2: also without the key fill (no LDS access at all)
walk_old = "    if (fmask[row]) {  // rows without a frame are zeroed below; skip their walk"
assert walk_old in text
e1 = text.replace(walk_old, "    if (false) {").replace(old, "    float r0 = qr[0] + 0.f * qd[0], r1 = qr[1] + 0.f * qd[1], r2 = qr[2] + 0.f * qd[2];\n    den = 0.f;\n    if (den > 0.f) {")
tmp.write_text(e1)
build("e1", asm_of(tmp))
fill_old = "  for (int l = lane; l < L; l += 64) {\n    const int64_t row = row0 + l;\n    float R[9], t[3], v[3], o[3];"
assert fill_old in e1
tmp.write_text(e1.replace(fill_old, "  for (int l = lane; l < 0; l += 64) {\n    const int64_t row = row0 + l;\n    float R[9], t[3], v[3], o[3];"))
build("e2", asm_of(tmp))
tmp.unlink()
