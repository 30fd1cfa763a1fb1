#!/usr/bin/env python3
"""Measure `roofline.traffic` of the dominant kernel (the FFN-up GEMM) and write the record bench.py reads.

    python tools/pmc_traffic.py [profiles/rNN_gemm_traffic.json]        (on the GPU box; ~1 min)

Separate rocprofv3 passes, counters only with --kernel-trace (gpurun refuses --pmc together with the hip / hsa trace domains):
FETCH_SIZE and WRITE_SIZE over three launches of g4::gemm256w4_kernel<SWIGLU> at the two row counts the forward issues
(M = 12 900: one sub-batch stream of configs[1]; M = 25 800: the whole batch), through scratch/one_gemm.py.  Corrections as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts half of a wide coalesced read stream -> x2; both
counters are in KB).  The figure is L2 -> fabric traffic: hits in the 256 MB Infinity Cache are included, so it bounds HBM bytes
from above (no counter behind that cache is exposed on this stack).

The record carries the sha256 of csrc/gemm256w4.hip; bench.py reports `traffic` only while that hash matches the source the
library was built from, and says "stale" otherwise (a constant read from a committed file goes stale silently).
"""
import csv
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNEL_SRC = ROOT / "esmdiff_amd" / "csrc" / "gemm256w4.hip"


def kernel_source_sha256() -> str:
    return hashlib.sha256(KERNEL_SRC.read_bytes()).hexdigest()


def one_pass(counter: str, rows: int) -> float:
    """Mean counter value per launch of the 256x256 GEMM kernel in one rocprofv3 --pmc pass."""
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        env = dict(os.environ, MM=str(rows), TMPDIR="/tmp")
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                        sys.executable, str(ROOT / "scratch" / "one_gemm.py")], cwd="/tmp", env=env, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = list(Path(d).rglob("*counter_collection.csv"))
        if not files:
            raise RuntimeError(f"rocprofv3 wrote no counter_collection.csv for {counter}")
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                if "gemm256w4" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    if not vals:
        raise RuntimeError(f"no gemm256w4 launch in the {counter} pass")
    return sum(vals) / len(vals)


def main():
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "gemm_traffic.json"
    N, K = 8192, 1536
    rec = {"kernel": "g4::gemm256w4_kernel<SWIGLU> N=8192 K=1536 (FFN-up), keyed by rows M per launch",
           "kernel_source": str(KERNEL_SRC.relative_to(ROOT)), "kernel_source_sha256": kernel_source_sha256(),
           "correction": "gfx950 rocprofv3: FETCH_SIZE x2 (wide coalesced read streams are counted at half), WRITE_SIZE x1; both in KB",
           "note": "L2 -> fabric requests: Infinity Cache hits included (an upper bound of HBM bytes). Separate --pmc passes, kernel-trace only, three launches each",
           "tool": "tools/pmc_traffic.py", "by_rows": {}}
    for rows in (25800, 12900):
        f, w = one_pass("FETCH_SIZE", rows), one_pass("WRITE_SIZE", rows)
        rd, wr = int(f * 1024 * 2), int(w * 1024)
        rec["by_rows"][str(rows)] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "read_bytes": rd, "write_bytes": wr,
                                     "traffic_bytes_per_launch": rd + wr,
                                     "algorithmic_bytes_per_launch": 2 * (rows * K + N * K) + 2 * rows * (N // 2)}
    out.parent.mkdir(exist_ok=True)
    out.write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec["by_rows"]))


if __name__ == "__main__":
    main()
