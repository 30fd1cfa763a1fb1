#!/bin/bash
# cache-policy variants of the four-wave GEMM (GPU box): bash scratch/policy_ab.sh
for v in "" ant wnt bnt asc1 wsc1 ""; do
  lib=esmdiff_amd/lib/libesmdiff_hip${v:+_$v}.so
  echo "== ${v:-product}"
  for m in 25800 12900; do
    M=$m ESMDIFF_LIB=$PWD/$lib python scratch/bench_gemm.py 2>&1 | grep -E "^qkv|^out |^ffn_up|^ffn_down |block" | awk -v m=$m '{printf "M=%s %s %s us %s TF/s | ", m, $1, $5, $7} END{print ""}'
  done
done
for v in "" bnt wnt ant ""; do
  lib=esmdiff_amd/lib/libesmdiff_hip${v:+_$v}.so
  ESMDIFF_LIB=$PWD/$lib python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-step0-sharing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ${v:-product}', d['value'], d['power']['mean_w'], d['power']['mean_sclk_mhz'])"
done
