#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_strict.py tests/test_gpu_fullwidth.py tests/test_gpu_kernels.py -m gpu -q -k "frames_is_batch_independent or coordinates or inpainting or geom" 2>&1 | tail -8 > gpurun_out/r06_t15.log
rm -f gpurun_out/r06_soak_gibbs_cfg4.txt
timeout 900 python tools/certified_soak.py --mode gibbs --weights random --inpaint --steps 50 --jobs 40 --first_seed 9000 --budget_s 700 --out gpurun_out/r06_soak_gibbs_cfg4.txt > /dev/null 2>&1
tail -6 gpurun_out/r06_t15.log; grep "^#" gpurun_out/r06_soak_gibbs_cfg4.txt | tail -7 | cut -c1-300
