#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 900 bash scratch/r06_attention_ab.sh > /dev/null 2>&1
rm -f gpurun_out/r06_gibbs_probe2.txt
for cfg in "--weights trained_like" "--weights trained_like --inpaint --steps 50" "--weights random"; do
  timeout 900 python tools/certified_soak.py --mode gibbs $cfg --jobs 3 --out gpurun_out/r06_gibbs_probe2.txt 2>&1 | tail -2
done
cat gpurun_out/r06_attention_packed_ab.txt
