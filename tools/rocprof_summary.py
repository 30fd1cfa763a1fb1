#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) into the plain-text per-kernel
summary committed under profiles/.   usage: tools/rocprof_summary.py <results.db> [<out.txt>] [--skip-first N]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void ", "").replace("ed::", "")
    return name[:90]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    db = sqlite3.connect(args[0])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
             f"{'vgpr':>5s} {'agpr':>5s} {'lds':>7s} {'grid':>9s} {'wg':>5s}"]
    for r in rows:
        lines.append(f"{short(r[0]):90s} {r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:9.2f} "
                     f"{r[5] / 1e3:9.2f} {100 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:7d} {r[9]:9d} {r[10]:5d}")
    lines.append(f"{'TOTAL kernel time':90s} {sum(r[1] for r in rows):7d} {total / 1e6:10.3f}")
    text = "\n".join(lines) + "\n"
    if len(args) > 1:
        open(args[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
