#!/bin/bash
# samples/s (ms per batch) over the batch size at three lengths, default settings (GPU box): bash scratch/batch_sweep_all.sh
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
run() { python $R/bench.py --residues $1 --samples-per-gpu $2 --steps 4 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('L_res=$1 B=$2', d['value'], 'samples/s', d['ms_per_step'], 'ms/batch')"; }
for b in 4 8 12 16 18 24 32 40 48 64; do run 58 $b; done
for b in 4 8 10 12 16 20 24; do run 126 $b; done
for b in 3 4 5 6 7 8 10 12; do run 256 $b; done
cd $R
