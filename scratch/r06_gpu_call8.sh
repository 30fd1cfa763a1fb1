#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
rm -f gpurun_out/r06_diag_*.txt
timeout 500 python tools/certified_soak.py --mode gibbs --weights random --inpaint --steps 50 --jobs 12 --first_seed 9000 --direct_share 1.0 --out gpurun_out/r06_diag_nodirect.txt > /dev/null 2>&1
timeout 700 python tools/certified_soak.py --mode gibbs --weights random --inpaint --steps 50 --jobs 12 --first_seed 9000 --audit_rate 1.0 --out gpurun_out/r06_diag_audit_all.txt > /dev/null 2>&1
grep "^#   seed\|^# jobs\|^# audits" gpurun_out/r06_diag_nodirect.txt gpurun_out/r06_diag_audit_all.txt
