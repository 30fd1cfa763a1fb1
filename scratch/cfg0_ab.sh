#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
run() { echo "== SPG=${SPG:-4} RES=${RES:-58} $*"; env "$@" python $R/bench.py --residues ${RES:-58} --samples-per-gpu ${SPG:-4} --steps ${ST:-20} --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'samples/s', d['ms_per_step'], 'ms/batch')"; }
run ESMDIFF_SMALL_ATTN_FUSED=0
run ESMDIFF_SMALL_ATTN_FUSED=1
ST=6 RES=256 SPG=3 run ESMDIFF_SMALL_ATTN_FUSED=0
ST=6 RES=256 SPG=3 run ESMDIFF_SMALL_ATTN_FUSED=1
cd $R
