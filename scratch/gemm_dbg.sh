#!/bin/bash
# Ablations of the 256x256 GEMM main loop (needs a -DED_GEMM_DEBUG build): which wait does the K-tile spend its time in?
# ESMDIFF_GEMM_DBG bits: 2 no LDS-DMA in the main loop, 4 no stores, 8 no vmcnt wait, 16 no barrier, 1 no fragment reads,
# 64 no MFMA, 128 no lgkmcnt waits.  Results are wrong by construction; only the times mean something.
ESMDIFF_EXTRA_CXXFLAGS=-DED_GEMM_DEBUG python -m esmdiff_amd.build --force > /dev/null 2>&1
for d in 0 8 16 24 4 12 28 2 1 129; do
  echo "== ESMDIFF_GEMM_DBG=$d"
  ESMDIFF_GEMM_DBG=$d python scratch/bench_gemm.py 2>&1 | grep -E "qkv|^out |ffn_up|ffn_down |block"
done
python -m esmdiff_amd.build --force > /dev/null 2>&1
