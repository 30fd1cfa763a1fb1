#!/bin/bash
# GPU power / clock samples while the bench runs (GPU box): bash scratch/power_trace.sh [bench args...]
out=$PWD/gpurun_out/power_trace.txt
rocm-smi --showmaxpower 2>&1 | grep -i "power (W)" > $out
python bench.py --steps 10 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/power_bench.json 2>/dev/null &
pid=$!
t0=$(date +%s.%N)
while kill -0 $pid 2>/dev/null; do
  s=$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ')
  echo "$(echo "$(date +%s.%N) - $t0" | bc | cut -c1-6) $s" >> $out
done
wait $pid
cut -c1-120 gpurun_out/power_bench.json >> $out
