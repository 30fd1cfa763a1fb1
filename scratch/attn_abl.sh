#!/bin/bash
for v in "$@"; do
  lib=esmdiff_amd/lib/libesmdiff_hip${v:+_$v}.so
  echo "== ${v:-product}"
  ESMDIFF_LIB=$PWD/$lib python scratch/time_attention.py 2>&1 | grep "us per"
done
