#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 1500 bash scratch/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
( time timeout 600 python bench.py > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err ) 2> gpurun_out/r06_default_bench.time
ls -la gpurun_out | grep r06_ | awk '{print $5, $9}'
python - <<'PY'
import json
for f in ("gpurun_out/r06_bench.json", "gpurun_out/r06_default_bench.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), "cert", d["certified"]["value"], d["certified"]["streamed"]["value"], d["certified"]["ids_equal_to_f32_split_chain"],
              "gibbs", {k: (v["value"], v["ratio_to_f32_split"], v["ids_equal_to_f32_split_chain"]) for k, v in d.get("certified_gibbs", {}).items() if isinstance(v, dict)}, "cpu", d.get("cpu_baseline", {}).get("value"))
    except Exception as ex:
        print(f, "ERR", ex)
PY
cat gpurun_out/r06_default_bench.time | tail -3
