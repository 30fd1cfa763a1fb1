#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -f gpurun_out/r06_soak_ddpm_trained_like.txt gpurun_out/r06_soak_gibbs_cfg1.txt gpurun_out/r06_soak_gibbs_cfg4.txt
timeout 1100 python tools/certified_soak.py --mode ddpm --weights trained_like --jobs 100 --first_seed 7000 --budget_s 900 --out gpurun_out/r06_soak_ddpm_trained_like.txt > /dev/null 2>&1
timeout 500 python tools/certified_soak.py --mode gibbs --weights random --jobs 30 --first_seed 8000 --budget_s 380 --out gpurun_out/r06_soak_gibbs_cfg1.txt > /dev/null 2>&1
timeout 700 python tools/certified_soak.py --mode gibbs --weights random --inpaint --steps 50 --jobs 30 --first_seed 9000 --budget_s 560 --out gpurun_out/r06_soak_gibbs_cfg4.txt > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_strict.py -m gpu -q -k "certified_gibbs_equals" 2>&1 | tail -15 > gpurun_out/r06_t7.log
tail -4 gpurun_out/r06_soak_ddpm_trained_like.txt; tail -4 gpurun_out/r06_soak_gibbs_cfg1.txt; tail -4 gpurun_out/r06_soak_gibbs_cfg4.txt; tail -5 gpurun_out/r06_t7.log
