#!/bin/bash
# bash scratch/ubench/mfma_power.sh  (GPU box): the MFMA-only ceiling under the power cap + rocm-smi samples beside it
out=$PWD/gpurun_out/mfma_power.txt
./scratch/ubench/mfma_power 4 > $out.run 2>&1 &
pid=$!
: > $out.smi
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ' >> $out.smi
  echo >> $out.smi
done
wait $pid
cat $out.run > $out
echo "--- rocm-smi samples (sclk, W), every 10th ---" >> $out
awk 'NR%10==0' $out.smi >> $out
