"""r06: ISA-level variants of the reduced victim (scratch/ubench/pk_rotate.hip, rotate_kernel<64>): hipcc's assembly with chosen packed
float ops rewritten as two plain float ops (through two spare registers, so that operand aliasing cannot matter), assembled into
rot_<tag>.co for scratch/r06_pk_rotate.py (CONFIGS=co:<tag>:64:1:12480).  Which packed ops have to go for the fault to disappear?"""
import re, subprocess, sys
from pathlib import Path
HERE = Path(__file__).resolve().parent
LLVM = "/opt/rocm/lib/llvm/bin/"
SRC = HERE.parent / "pk_rotate.hip"
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-fast-math", "-S", "--cuda-device-only", str(SRC), "-o", "/dev/stdout"],
                   capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-2000:]
base = r.stdout
T0, T1 = 60, 61          # spare registers (the kernel uses 50)
base = base.replace(".amdhsa_next_free_vgpr 50", ".amdhsa_next_free_vgpr 62").replace(".amdhsa_accum_offset 52", ".amdhsa_accum_offset 64").replace(".vgpr_count:     50", ".vgpr_count:     62")
PK = re.compile(r"^\tv_pk_(mul|add|fma)_f32 v\[(\d+):(\d+)\], ([^\n]*)$", re.M)


def operand(tok, half_hi):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return f"v{int(m.group(2)) if half_hi else int(m.group(1))}"
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return f"s{int(m.group(2)) if half_hi else int(m.group(1))}"
    return tok           # inline constant


def scalarise(m):
    op, dlo, dhi, rest = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
    mods = {"op_sel": None, "op_sel_hi": None}
    for k in mods:
        mm = re.search(rf"\b{k}:\[([01,]+)\]", rest)
        if mm:
            mods[k] = [int(x) for x in mm.group(1).split(",")]
    assert "neg" not in rest, rest
    ops = [t for t in re.sub(r"\bop_sel(_hi)?:\[[01,]+\]", "", rest).split(",") if t.strip()]
    n = len(ops)
    sel = mods["op_sel"] or [0] * n
    selh = mods["op_sel_hi"] or [1] * n
    lo = [operand(t, sel[i]) for i, t in enumerate(ops)]
    hi = [operand(t, selh[i]) for i, t in enumerate(ops)]
    ins = {"mul": "v_mul_f32_e64", "add": "v_add_f32_e64", "fma": "v_fma_f32"}[op]
    return (f"\t{ins} v{T0}, {', '.join(lo)}\n\t{ins} v{T1}, {', '.join(hi)}\n\tv_mov_b32_e32 v{dlo}, v{T0}\n\tv_mov_b32_e32 v{dhi}, v{T1}")


def build(tag, pick):
    k = 0
    def f(m):
        nonlocal k
        k += 1
        return scalarise(m) if pick(k, m.group(0)) else m.group(0)
    # only inside rotate_kernel<64>
    i = base.index("_Z13rotate_kernelILi64EE")
    j = base.index("s_endpgm", i)
    text = base[:i] + PK.sub(f, base[i:j]) + base[j:]
    (HERE / f"rot_{tag}.s").write_text(text)
    subprocess.run([LLVM + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(HERE / f"rot_{tag}.s"), "-o", str(HERE / f"rot_{tag}.o")], check=True)
    subprocess.run([LLVM + "ld.lld", "-shared", str(HERE / f"rot_{tag}.o"), "-o", str(HERE / f"rot_{tag}.co")], check=True)
    left = len(PK.findall(text[i:text.index("s_endpgm", i)]))
    print(tag, "built;", k, "packed mul/add/fma in the kernel,", left, "left packed")


i = base.index("_Z13rotate_kernelILi64EE"); j = base.index("s_endpgm", i)
for n, m in enumerate(PK.finditer(base[i:j]), 1):
    print(n, m.group(0).strip())
build("r0", lambda k, s: False)
build("rall", lambda k, s: True)
build("ropsel", lambda k, s: "op_sel" in s)
build("rplain", lambda k, s: "op_sel" not in s)
for n in range(1, 14):
    build(f"only{n}", lambda k, s, n=n: k != n)     # everything scalar except packed op n


# ---- the failing instruction (#6 reads v37 through op_sel; two instructions later v_mov_b32 v37, v42 overwrites v37) --------------
def build_text(tag, text):
    (HERE / f"rot_{tag}.s").write_text(text)
    subprocess.run([LLVM + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(HERE / f"rot_{tag}.s"), "-o", str(HERE / f"rot_{tag}.o")], check=True)
    subprocess.run([LLVM + "ld.lld", "-shared", str(HERE / f"rot_{tag}.o"), "-o", str(HERE / f"rot_{tag}.co")], check=True)
    print(tag, "built")


only6 = (HERE / "rot_only6.s").read_text()
i6 = only6.index("_Z13rotate_kernelILi64EE")
six = "\tv_pk_fma_f32 v[34:35], v[44:45], v[36:37], v[34:35] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
war = "\tv_mov_b32_e32 v37, v42\n"
p6 = only6.index(six, i6)
assert only6.index(war, p6) - p6 < 200
for tag, pad in (("w_nop1", "\ts_nop 0\n"), ("w_nop2", "\ts_nop 1\n"), ("w_nop4", "\ts_nop 3\n"), ("w_nop8", "\ts_nop 7\n"), ("w_nop16", "\ts_nop 7\n\ts_nop 7\n"), ("w_nop64", "\ts_nop 7\n" * 8)):
    build_text(tag, only6[:p6] + six + pad + only6[p6 + len(six):])          # idle cycles between #6 and what follows
# the overwrite of v37 goes to a spare register instead (v62), and its readers behind it read v62
tail = only6[p6 + len(six):]
j = tail.index("s_branch")
body = tail[:j].replace("v_mov_b32_e32 v37, v42", "v_mov_b32_e32 v62, v42").replace("v_fma_f32 v61, v31, v37, v41", "v_fma_f32 v61, v31, v62, v41")
assert "v37" not in body.split("v_mov_b32_e32 v62, v42")[1], body
build_text("w_rename", (only6[:p6] + six + body + tail[j:]).replace(".amdhsa_next_free_vgpr 62", ".amdhsa_next_free_vgpr 63"))
# all packed (the compiler's code) with 16 idle cycles behind #6 only
r0 = (HERE / "rot_r0.s").read_text()
q = r0.index(six, r0.index("_Z13rotate_kernelILi64EE"))
build_text("r0_nop16", r0[:q] + six + "\ts_nop 7\n\ts_nop 7\n" + r0[q + len(six):])
