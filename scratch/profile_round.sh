#!/bin/bash
# End-of-phase measurement records (run on the GPU box):  bash scratch/profile_round.sh r02c
# 1. the default bench line (configs[1], with cpu_baseline + parity_spot)          -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace of configs[1], [3], [4]                               -> gpurun_out/<tag>_cfg{1,3,4}_kernel_stats.txt
# 3. PMC passes over three launches of the dominant kernel (FFN-up GEMM), one counter group per run, kernel-trace only:
#    SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE at M = 25800 and 12900 -> gpurun_out/<tag>_pmc_*.txt
tag=${1:-r02}
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
python $R/bench.py --steps 5 --warmup 1 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
prof() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name --output-format rocpd -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/${tag}_${name}_prof_bench.json 2> $out/${tag}_${name}_prof.err
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $db $out/${tag}_${name}_kernel_stats.txt --lib $R/esmdiff_amd/lib/libesmdiff_hip.so --union "gemm256w4_kernel<2, 0>" > /dev/null
}
# (the profiled runs skip the labelled extra legs: one engine, the headline workload only)
prof cfg1 --no-head-f32-leg --no-step0-sharing --no-breakdown
prof cfg3 --residues 1024 --samples-per-gpu 32 --no-head-f32-leg --no-step0-sharing
prof cfg4 --num-steps 50 --inpaint 96:160 --no-head-f32-leg --no-step0-sharing
prof cfg1_f16 --precision f16 --no-head-f32-leg --no-step0-sharing --no-breakdown
prof cfg1_f32split --precision f32_split --steps 1 --warmup 1 --no-step0-sharing
for mm in 25800 12900; do
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/pmc_$n
    MM=$mm rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$n -o pmc --output-format csv -- python $R/scratch/one_gemm.py > /dev/null 2>&1
    f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
    echo "== M=$mm $grp" >> $out/${tag}_pmc_gemm.txt
    python $R/scratch/pmc_rows.py $f gemm256 >> $out/${tag}_pmc_gemm.txt
  done
done
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do   # the F32_SPLIT FFN-up (three f16 passes)
  n=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pmcs_$n
  MM=25800 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcs_$n -o pmc --output-format csv -- python $R/scratch/one_gemm_split.py > /dev/null 2>&1
  f=$(find /tmp/pmcs_$n -name "*counter_collection.csv" | head -1)
  echo "== split FFN-up M=25800 $grp" >> $out/${tag}_pmc_gemm_split.txt
  python $R/scratch/pmc_rows.py $f gemm256 >> $out/${tag}_pmc_gemm_split.txt
done
cd $R
