#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
for bm in 64 128; do
rm -rf /tmp/ks; ESMDIFF_GEMM_SMALL_BM=$bm rocprofv3 --kernel-trace -d /tmp/ks -o ks --output-format rocpd -- python $R/scratch/small_gemm_ksweep.py > /dev/null 2>&1
db=$(find /tmp/ks -name "*.db" | head -1)
echo "== BM=$bm (K = 128 256 512 1024 1536 3072; median us of 50 launches each)"
python - $db <<'PY'
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, duration from kernels where name like '%gemm_bf16_kernel%' order by start").fetchall()
d = [r[1] / 1e3 for r in rows]
for i in range(0, len(d), 50):
    print(round(statistics.median(d[i:i + 50]), 2), end="  ")
print()
PY
done
cd $R
