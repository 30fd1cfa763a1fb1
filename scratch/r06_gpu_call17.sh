#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_strict.py -m gpu -q -k "frames_is_batch_independent" 2>&1 | tail -3
rm -f gpurun_out/r06_soak_gibbs_tl_cfg4.txt gpurun_out/r06_soak_gibbs_tl_cfg1.txt
timeout 700 python tools/certified_soak.py --mode gibbs --weights trained_like --inpaint --steps 50 --jobs 20 --first_seed 11000 --budget_s 560 --out gpurun_out/r06_soak_gibbs_tl_cfg4.txt > /dev/null 2>&1
timeout 300 python tools/certified_soak.py --mode gibbs --weights trained_like --jobs 10 --first_seed 12000 --budget_s 200 --out gpurun_out/r06_soak_gibbs_tl_cfg1.txt > /dev/null 2>&1
grep "^#" gpurun_out/r06_soak_gibbs_tl_cfg4.txt | tail -6 | cut -c1-260; grep "^#" gpurun_out/r06_soak_gibbs_tl_cfg1.txt | tail -6 | cut -c1-260
