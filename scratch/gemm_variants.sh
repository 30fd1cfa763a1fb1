#!/bin/bash
# usage (GPU box): bash scratch/gemm_variants.sh "" abl8 abl16 ...   ("" = the product library)
for v in "$@"; do
  lib=esmdiff_amd/lib/libesmdiff_hip${v:+_$v}.so
  echo "== ${v:-product}"
  ESMDIFF_LIB=$PWD/$lib python scratch/bench_gemm.py 2>&1 | grep -E "qkv|^out |ffn_up|ffn_down |block"
done
