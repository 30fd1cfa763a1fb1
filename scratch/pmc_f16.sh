#!/bin/bash
# PMC of the FFN-up GEMM in its f16 build next to the bf16 build (same shape, alone on the GPU): MFMA-busy cycles and the clock
# the part sustains under each -> gpurun_out/<tag>_pmc_gemm_f16.txt
tag=${1:-r04}
out=$PWD/gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for s in one_gemm one_gemm_f16; do
  rm -rf /tmp/pmcf_$s
  MM=25800 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmcf_$s -o pmc --output-format csv -- python $R/scratch/$s.py > /dev/null 2>&1
  f=$(find /tmp/pmcf_$s -name "*counter_collection.csv" | head -1)
  echo "== $s M=25800 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" >> $out/${tag}_pmc_gemm_f16.txt
  python $R/scratch/pmc_rows.py $f gemm256 >> $out/${tag}_pmc_gemm_f16.txt
done
cd $R
