#!/bin/bash
run() { env $1 python bench.py --samples-per-gpu $2 --steps 4 --warmup 1 --no-cpu-baseline --no-profile --no-step0-sharing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 B=$2', d['value'])"; }
for b in 10 16 32; do
  for ds in 1 2 3; do
    for mt in 96 128 192; do
      run "ESMDIFF_DUAL_STREAM=$ds ESMDIFF_GEMM_256_MIN_TILES=$mt" $b
    done
  done
done
