#!/bin/bash
# r04: A/B of the TINY key-tail path of the attention kernel (<= 4 valid keys in the last tile: L_tok = 258) in one box session:
# rope + attention alone on the GPU at configs[1] / configs[3] shapes (B = 50 is the two-stream launch size), 3 rounds each,
# then the bench's per-section attention time and samples/s.  -> gpurun_out/r04_attention_tiny_tail_ab.txt
out=$PWD/gpurun_out/r04_attention_tiny_tail_ab.txt
: > $out
for r in 1 2 3; do
  for t in 0 1; do
    echo "== round $r ESMDIFF_ATTN_TINY_TAIL=$t" >> $out
    ESMDIFF_ATTN_TINY_TAIL=$t python scratch/time_attention.py 2>/dev/null >> $out
  done
done
for t in 0 1 0 1; do
  ESMDIFF_ATTN_TINY_TAIL=$t python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-head-f32-leg --no-step0-sharing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench ESMDIFF_ATTN_TINY_TAIL=$t', d['value'], 'samples/s; attention', d['sections_ms_per_forward']['attention'], 'ms per forward')" >> $out
done
cat $out
