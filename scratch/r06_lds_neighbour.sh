#!/bin/bash
# r06: positive control (the hazard on an A/B build of the geometric kernel WITH packed float ops — the product build has none:
#   python scratch/build_variant.py geom geompk -fslp-vectorize -fvectorize      [here, before the gpurun call]
# ), the product build on the same inputs, then the LDS-integrity probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
  echo "=== geompk build (packed float ops): rows hit (one block) ==="
  ESMDIFF_LIB=esmdiff_amd/lib/libesmdiff_hip_geompk.so timeout 600 python scratch/r06_frames_race_rows.py
  echo "=== product build (no packed float ops): the same ==="
  FORWARDS=4 timeout 600 python scratch/r06_frames_race_rows.py
  echo "=== neighbour probe ==="
  timeout 900 python scratch/r06_lds_neighbour.py
} > gpurun_out/r06_lds_neighbour.txt 2>&1
tail -40 gpurun_out/r06_lds_neighbour.txt
