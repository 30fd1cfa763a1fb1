#!/bin/bash
# kernel-trace summary of one mid-size batch (GPU box): bash scratch/profile_b.sh <B> [residues]
B=${1:-8}; RES=${2:-256}
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o pb --output-format rocpd -- python $R/bench.py --residues $RES --samples-per-gpu $B --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $out/pb_${B}_bench.json 2> /dev/null
db=$(find /tmp/prof_b -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $out/pb_${B}_kernel_stats.txt > /dev/null
cd $R
head -12 $out/pb_${B}_kernel_stats.txt | cut -c1-160
