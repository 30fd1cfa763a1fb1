#!/bin/bash
# r04: one vs two sub-batch streams at L_tok 258 for the batch sizes around the reference's default --num_samples 10
# (sample_esmdiff.py:243) -> gpurun_out/r04_mid_streams.txt   (samples/s; same box session)
out=$PWD/gpurun_out/r04_mid_streams.txt
: > $out
for B in 8 9 10 11 12 13 14 16; do
  for ds in 1 2; do
    v=$(ESMDIFF_DUAL_STREAM=$ds python bench.py --samples-per-gpu $B --steps 4 --warmup 1 --no-cpu-baseline --no-head-f32-leg --no-step0-sharing --no-breakdown 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
    echo "B=$B ESMDIFF_DUAL_STREAM=$ds $v" >> $out
  done
done
cat $out
