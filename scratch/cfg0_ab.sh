#!/bin/bash
# configs[0] A/B of the small-batch path knobs (GPU box): bash scratch/cfg0_ab.sh
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
run() { echo "== SPG=${SPG:-4} $*"; env "$@" python $R/bench.py --residues 58 --samples-per-gpu ${SPG:-4} --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'samples/s', d['ms_per_step'], 'ms/batch')"; }
run ESMDIFF_SMALL_FUSED=1
run ESMDIFF_SMALL_FUSED=1 ESMDIFF_GEMM_SMALL_STAGES=6
SPG=8 run ESMDIFF_SMALL_FUSED=1
SPG=8 run ESMDIFF_SMALL_FUSED=1 ESMDIFF_GEMM_SMALL_STAGES=6
cd $R
