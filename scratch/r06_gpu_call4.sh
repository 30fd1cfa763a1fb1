#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gibbs_rows or gibbs_margin" 2>&1 | tail -30 > gpurun_out/r06_t4.log
for cfg in "--weights random" "--weights trained_like" "--weights random --inpaint --steps 50" "--weights trained_like --inpaint --steps 50"; do
  timeout 900 python tools/certified_soak.py --mode gibbs $cfg --jobs 3 --out gpurun_out/r06_gibbs_probe.txt 2>&1 | tail -3
done
timeout 1500 bash scratch/r06_raster_ab.sh > /dev/null 2>&1
tail -5 gpurun_out/r06_t4.log
