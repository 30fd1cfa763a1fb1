#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_strict.py -m gpu -q -k "frames_is_batch_independent" 2>&1 | tail -15 > gpurun_out/r06_t11.log
rm -f gpurun_out/r06_soak_gibbs_cfg4.txt
timeout 800 python tools/certified_soak.py --mode gibbs --weights random --inpaint --steps 50 --jobs 30 --first_seed 9000 --budget_s 640 --out gpurun_out/r06_soak_gibbs_cfg4.txt > /dev/null 2>&1
tail -6 gpurun_out/r06_t11.log; grep "^#" gpurun_out/r06_soak_gibbs_cfg4.txt | tail -8 | cut -c1-300
