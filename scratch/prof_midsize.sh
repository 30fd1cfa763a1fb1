#!/bin/bash
# kernel-trace summaries of the mid-size batches (GPU box): bash scratch/prof_midsize.sh [B ...] -> gpurun_out/mid_B*_kernel_stats.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for B in ${@:-10 16 32}; do
  rm -rf /tmp/prof_mid
  rocprofv3 --kernel-trace --stats -d /tmp/prof_mid -o mid --output-format rocpd -- python $R/bench.py --samples-per-gpu $B --steps 3 --warmup 1 --no-cpu-baseline --no-step0-sharing > $R/gpurun_out/mid_B${B}_bench.json 2> $R/gpurun_out/mid_B${B}.err
  db=$(find /tmp/prof_mid -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $db $R/gpurun_out/mid_B${B}_kernel_stats.txt > /dev/null
done
cd $R
