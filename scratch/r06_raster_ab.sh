#!/bin/bash
# r06: tile-raster A/B of gemm256w4 (run on the GPU box; the variant libraries are built on the CPU container by
#   python scratch/build_variant.py gemm256w4 gm4 -DED_W4_GROUP_M=4           etc. — see EXPERIMENTS.md R6.3)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r06_gemm_raster_ab.txt
: > $OUT
for tag in "" gm4 gm4ntw r1g4 r1g4nta; do
  lib=esmdiff_amd/lib/libesmdiff_hip${tag:+_$tag}.so
  [ -f "$lib" ] || continue
  echo "== variant ${tag:-base} ($lib)" >> $OUT
  ESMDIFF_LIB=$PWD/$lib timeout 600 python scratch/r06_raster_ab.py 2>&1 | tail -1 >> $OUT
  ESMDIFF_LIB=$PWD/$lib timeout 600 python tools/pmc_traffic.py gpurun_out/r06_traffic_${tag:-base}.json 2>&1 | tail -1 >> $OUT
done
cat $OUT
