#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for v in $VARIANTS; do
  echo "=== variant $v ==="
  ESMDIFF_LIB=esmdiff_amd/lib/libesmdiff_hip_$v.so PRECS=${PRECS:-bf16} FORWARDS=${FORWARDS:-6} timeout 300 python scratch/r06_frames_race_rows.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r06_probe_$TAG.txt 2>&1
cat gpurun_out/r06_probe_$TAG.txt | cut -c1-700
