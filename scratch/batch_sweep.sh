#!/bin/bash
# samples/s over the batch size at L_tok = 258 (GPU box): bash scratch/batch_sweep.sh
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
for spg in 1 3 8 12 16 20 24 28 32 36 40 44 48 64 100; do
  python $R/bench.py --residues 256 --samples-per-gpu $spg --steps 4 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$spg', d['value'], 'samples/s', d['ms_per_step'], 'ms/batch', d['power']['mean_w'] if d.get('power') else None, 'W')"
done
cd $R
