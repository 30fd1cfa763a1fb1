#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_strict.py tests/test_gpu_fullwidth.py -m gpu -q -k "certified or gibbs or stream_options or whole_configs1 or logit_error" 2>&1 | tail -60 > gpurun_out/r06_t6.log
cp gpurun_out/parity_strict.json gpurun_out/r06_parity_strict_partial.json 2>/dev/null
rm -f gpurun_out/r06_soak_ddpm_random.txt
timeout 2100 python tools/certified_soak.py --mode ddpm --weights random --jobs 200 --first_seed 5000 --budget_s 1750 --out gpurun_out/r06_soak_ddpm_random.txt > /dev/null 2>&1
tail -8 gpurun_out/r06_t6.log; tail -7 gpurun_out/r06_soak_ddpm_random.txt
