#!/bin/bash
# configs[0] (B 4, L_tok 60): bench line + kernel-trace summary (GPU box):  bash scratch/profile_cfg0.sh <tag>
tag=${1:-r02}
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
python $R/bench.py --residues 58 --samples-per-gpu 4 --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_cfg0_bench.json 2> $out/${tag}_cfg0_bench.err
rm -rf /tmp/prof_cfg0
rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg0 -o cfg0 --output-format rocpd -- python $R/bench.py --residues 58 --samples-per-gpu 4 --steps 5 --warmup 1 --no-cpu-baseline --no-profile > $out/${tag}_cfg0_prof_bench.json 2> $out/${tag}_cfg0_prof.err
db=$(find /tmp/prof_cfg0 -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $db $out/${tag}_cfg0_kernel_stats.txt > /dev/null
cd $R
