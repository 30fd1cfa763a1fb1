#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r06_smoke.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06_t16.log
cp gpurun_out/parity_strict.json gpurun_out/r06_parity_strict.json 2>/dev/null
cp gpurun_out/parity_fullwidth.json gpurun_out/r06_parity_fullwidth.json 2>/dev/null
timeout 700 python bench.py --steps 5 --warmup 1 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
( time timeout 700 python bench.py > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err ) 2> gpurun_out/r06_default_bench.time
cat gpurun_out/r06_smoke.log | cut -c1-600; tail -4 gpurun_out/r06_t16.log
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench.json", "gpurun_out/r06_default_bench.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["roofline"]["frac"], "cert", d["certified"]["value"], d["certified"]["streamed"]["value"], d["certified"]["ids_equal_to_f32_split_chain"],
              "gibbs", {k: (v["value"], v["ratio_to_f32_split"], v["ids_equal_to_f32_split_chain"]) for k, v in d.get("certified_gibbs", {}).items() if isinstance(v, dict)}, "cpu", d.get("cpu_baseline", {}).get("value"))
    except Exception as ex:
        print(f, "ERR", ex)
PY
tail -3 gpurun_out/r06_default_bench.time
