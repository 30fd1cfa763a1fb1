#!/bin/bash
# PMC passes over the HBM-bound / latency-bound kernels of configs[1] (GPU box): bash scratch/pmc_small_kernels.sh <tag>
# one counter group per run, kernel-trace only (never combined with sys/hip traces)
tag=${1:-r02}
out=$PWD/gpurun_out/${tag}_pmc_small_kernels.txt
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
: > $out
run() {  # script, kernel pattern, counters...
  script=$1; pat=$2; shift 2
  rm -rf /tmp/pmcs
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmcs -o pmc --output-format csv -- python $R/scratch/$script > /dev/null 2>&1
  f=$(find /tmp/pmcs -name "*counter_collection.csv" | head -1)
  echo "== $script [$pat] $*" >> $out
  python $R/scratch/pmc_rows.py $f $pat >> $out
}
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  run one_attention.py attention_kernel $grp
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  run one_attention.py qk_norm_rope $grp
  run one_ln.py layernorm $grp
done
cd $R
