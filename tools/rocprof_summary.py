#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) into the plain-text per-kernel summary committed
under profiles/.

    tools/rocprof_summary.py <results.db> [<out.txt>] [--lib esmdiff_amd/lib/libesmdiff_hip.so] [--union <kernel substring>]

--lib     register columns from the CODE OBJECT metadata of that library (llvm-objdump --offloading + llvm-readelf --notes):
          `vgpr` = architectural VGPRs, `agpr` = accumulation VGPRs.  rocprofv3's own columns are not usable on gfx950: it
          reports vgpr_count = (arch + acc) / 2 and accum_vgpr_count = 0 for a kernel whose metadata says 152 + 256 (VERDICT r03
          item 9: the r01-r03 summaries show `agpr 0` for g4::gemm256w4_kernel for that reason).
--union   for the kernels whose name contains the substring: merge the begin / end stamps of all their launches (all streams,
          one device timeline) and print the UNION of the busy intervals, next to the plain sum and the launch count — the
          figure bench.py reports as roofline.union.union_busy_ms, reproducible from the committed trace.
"""
import re
import sqlite3
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")          # before the argument list is cut at the first "("
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void ", "").replace("ed16::", "f16:").replace("ed::", "")
    return name[:90]


def codeobj_registers(lib: str):
    """{short demangled kernel name: (arch_vgprs, agprs, sgprs, spills)} from the library's embedded gfx950 code objects."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        tmp = Path(d) / "lib.so"
        tmp.write_bytes(Path(lib).read_bytes())
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(tmp)], capture_output=True, text=True, check=True)
        for co in sorted(Path(d).glob("lib.so.*gfx950*")):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s*(?:-\s*)?\.(agpr_count|vgpr_count|sgpr_count|sgpr_spill_count|vgpr_spill_count|name):\s+(\S+)", line)
                if not m:
                    continue
                if line.lstrip().startswith("-") and cur.get("name"):
                    out[cur["name"]] = cur
                    cur = {}
                elif line.lstrip().startswith("-"):
                    cur = {}
                cur[m.group(1)] = m.group(2)
            if cur.get("name"):
                out[cur["name"]] = cur
    names = list(out)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
    res = {}
    for mangled, d_ in zip(names, dem):
        c = out[mangled]
        ag, vg = int(c.get("agpr_count", 0)), int(c.get("vgpr_count", 0))
        res[short(d_)] = (vg - ag, ag, int(c.get("sgpr_count", 0)), int(c.get("sgpr_spill_count", 0)) + int(c.get("vgpr_spill_count", 0)))
    return res


def union_ms(intervals):
    intervals = sorted(intervals)
    busy, lo, hi = 0, intervals[0][0], intervals[0][1]
    for a, b in intervals[1:]:
        if a > hi:
            busy += hi - lo
            lo, hi = a, b
        elif b > hi:
            hi = b
    return (busy + hi - lo) / 1e6


def main():
    argv = sys.argv[1:]
    opts = {}
    for flag in ("--lib", "--union", "--skip-first"):
        if flag in argv:
            i = argv.index(flag)
            opts[flag] = argv[i + 1]
            del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    db = sqlite3.connect(args[0])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    regs = codeobj_registers(opts["--lib"]) if "--lib" in opts else {}
    total = sum(r[2] for r in rows) or 1
    src = "code-object metadata" if regs else "rocprofv3 columns (unreliable on gfx950)"
    lines = [f"# register columns: {src}",
             f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
             f"{'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'grid':>9s} {'wg':>5s}"]
    for r in rows:
        vg, ag = r[6], r[7]
        key = short(r[0])
        if key in regs:
            vg, ag = regs[key][0], regs[key][1]
        lines.append(f"{key:90s} {r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} "
                     f"{r[5] / 1e3:9.2f} {100 * r[2] / total:6.2f} {vg:5d} {ag:5d} {r[8]:7d} {r[9]:9d} {r[10]:5d}")
    lines.append(f"{'TOTAL kernel time':90s} {sum(r[1] for r in rows):7d} {total / 1e6:10.3f}")
    if "--union" in opts:
        cols = [d[0] for d in db.execute("select * from kernels limit 1").description]
        if "start" not in cols or "end" not in cols:
            raise SystemExit(f"--union: the kernels view has no start / end columns ({cols})")
        for (name,) in db.execute("select distinct name from kernels where name like ?", (f"%{opts['--union']}%",)).fetchall():
            iv = db.execute('select start, "end" from kernels where name = ? order by start', (name,)).fetchall()
            tot = sum(b - a for a, b in iv) / 1e6
            lines.append(f"UNION {short(name)}: launches {len(iv)}, sum of durations {tot:.3f} ms, union of busy intervals "
                         f"{union_ms(iv):.3f} ms, mean launch {tot / len(iv) * 1e3:.2f} us")
    text = "\n".join(lines) + "\n"
    if len(args) > 1:
        open(args[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
