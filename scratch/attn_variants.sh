#!/bin/bash
# Attention workgroup-width / register-budget sweep at BASELINE configs[1] (one bench step each; per-section HIP events).
# usage (GPU box): bash scratch/attn_variants.sh > gpurun_out/attn_variants.txt
for v in "3 4" "3 0" "4 0" "4 5" "4 9" "4 4"; do
  set -- $v
  echo "== ESMDIFF_ATTN_OCC=$1 ESMDIFF_ATTN_WAVES=$2 (0 = auto)"
  ESMDIFF_ATTN_OCC=$1 ESMDIFF_ATTN_WAVES=$2 python bench.py --steps 1 --warmup 1 --no-cpu-baseline ${EXTRA} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['sections_ms_per_forward']
print('samples/s', d['value'], 'attention ms/fwd', s['attention'], 'device ms/fwd', d['device_ms_per_forward'])"
done
