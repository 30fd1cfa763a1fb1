#!/usr/bin/env python3
"""Static check of the hand-placed GEMM kernels' ISA for a hazard hipcc cannot see.

The main loops of csrc/gemm256w4.hip are `asm volatile` statements; hipcc allocates registers around them and, when a kernel
runs out of SGPRs, spills scalars into VGPR lanes and reloads them with v_readlane_b32 wherever they are needed next — also
directly in front of an inline-asm `global_load_lds_dwordx4 v, s[a:b]` that uses the reloaded pair as its scalar base.  gfx9
needs 5 wait states between a VALU write of an SGPR and a VMEM instruction that reads it; the compiler pads that for its own
instructions, not for the text inside an asm statement.  (r04: the first build of the split-f16 kernel faulted at multi-tile
launches for exactly this reason.)

    python tools/check_asm_hazards.py esmdiff_amd/csrc/gemm256w4.hip [more .hip files]

compiles each file to gfx950 assembly (device only) and reports, per kernel: SGPR / VGPR spill counts from the code-object
metadata, and every inline-asm VMEM instruction whose scalar operands were written by a VALU instruction (v_readlane_b32,
v_readfirstlane_b32, v_cmp... with an SGPR destination) fewer than 5 wait states earlier.  Exit status 1 if any hazard exists.
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "--cuda-device-only", "-S"]
VMEM = re.compile(r"^\s*(global_load|global_store|buffer_load|buffer_store|scratch_|flat_)")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def wait_states(ins):
    m = re.match(r"\s*s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def analyse(asm_text):
    """Yields (kernel, sgpr_spills, vgpr_spills, hazards[list of str])."""
    kernels, meta = {}, {}
    cur = None
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w+):\s", line)
        if m and ".amdhsa" not in line:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            if re.match(r"^\s*s_endpgm", line):
                cur = None
            else:
                kernels[cur].append(line)
    name = None
    pending = {}        # the metadata item of a kernel starts with "- .agpr_count", its .name comes later
    for line in asm_text.splitlines():
        if re.match(r"\s*-\s*\.agpr_count:", line):
            name = None
            pending = {}
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
            meta.setdefault(name, {}).update(pending)
            pending = {}
        m = re.match(r"\s*(?:-\s*)?\.(sgpr_spill_count|vgpr_spill_count|agpr_count|vgpr_count|sgpr_count):\s+(\d+)", line)
        if m and name:
            meta[name][m.group(1)] = int(m.group(2))
        elif m:
            pending[m.group(1)] = int(m.group(2))
    for k, lines in kernels.items():
        hazards = []
        in_asm = False
        recent = []        # (instruction text, set of SGPRs written by a VALU op) newest last, non-comment instructions only
        for ln in lines:
            t = ln.split(";")[0].rstrip() if not ln.strip().startswith(";;#") else ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t.strip() or t.strip().endswith(":") or t.strip().startswith("."):
                continue
            ins = t.strip()
            if in_asm and VMEM.match(ins):
                need = sregs(ins)
                ws = 0
                for prev, written in reversed(recent):
                    if ws >= 5:
                        break
                    if written & need:
                        hazards.append(f"{prev}  ->  [{ws} wait states]  ->  {ins}")
                        break
                    ws += wait_states(prev)
            written = set()
            if re.match(r"v_(readlane|readfirstlane)_b32", ins) or (ins.startswith("v_cmp") and "_e64" in ins):
                written = sregs(ins.split(",")[0])
            recent.append((ins, written))
            if len(recent) > 12:
                recent.pop(0)
        mm = meta.get(k, {})
        yield k, mm, hazards


def compile_to_asm(src: Path) -> str:
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "k.s"
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, str(src), "-o", str(out)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        return out.read_text()


def main(argv):
    bad = 0
    for f in argv or ["esmdiff_amd/csrc/gemm256w4.hip"]:
        p = Path(f)
        text = p.read_text() if p.suffix == ".s" else compile_to_asm(p)
        for k, mm, hz in analyse(text):
            print(f"{k}: sgpr_spills {mm.get('sgpr_spill_count', '?')} vgpr_spills {mm.get('vgpr_spill_count', '?')} "
                  f"vgpr {mm.get('vgpr_count', '?')} agpr {mm.get('agpr_count', '?')} hazards {len(hz)}")
            for h in hz:
                print("    " + h)
            bad += len(hz)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
