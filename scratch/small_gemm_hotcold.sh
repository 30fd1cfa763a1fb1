#!/bin/bash
# bash scratch/small_gemm_hotcold.sh  (GPU box) -> per-kernel avg durations hot vs cold
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
for shape in qkv ffn_up; do for mode in hot cold; do
  rm -rf /tmp/hc; SHAPE=$shape MODE=$mode rocprofv3 --kernel-trace --stats -d /tmp/hc -o hc --output-format rocpd -- python $R/scratch/small_gemm_hotcold.py > /dev/null 2>&1
  db=$(find /tmp/hc -name "*.db" | head -1)
  echo "== $shape $mode $*"; python $R/tools/rocprof_summary.py $db /tmp/hc_stats.txt > /dev/null; grep gemm_bf16_kernel /tmp/hc_stats.txt | cut -c1-150
done; done
cd $R
