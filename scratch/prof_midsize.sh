#!/bin/bash
# kernel-trace summaries of the mid-size batches (GPU box): bash scratch/prof_midsize.sh  -> gpurun_out/mid_B*_stats.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for B in 10 16 32; do
  rm -rf /tmp/prof_mid
  rocprofv3 --kernel-trace --stats -d /tmp/prof_mid -o mid -- python $R/bench.py --samples-per-gpu $B --steps 3 --warmup 1 --no-cpu-baseline --no-step0-sharing > $R/gpurun_out/mid_B${B}_bench.json 2> $R/gpurun_out/mid_B${B}.err
  f=$(find /tmp/prof_mid -name "*kernel_stats.csv" | head -1)
  python - "$f" > $R/gpurun_out/mid_B${B}_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':110s} {'calls':>8s} {'avg_us':>9s} {'total_ms':>9s} {'%':>6s}")
for r in rows[:28]:
    print(f"{r['Name'][:110]:110s} {r['Calls']:>8s} {float(r['AverageNs'])/1e3:9.2f} {float(r['TotalDurationNs'])/1e6:9.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
PY
done
cd $R
