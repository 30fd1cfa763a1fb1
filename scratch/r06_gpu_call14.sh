#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 700 python bench.py --steps 5 --warmup 1 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
( time timeout 700 python bench.py > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err ) 2> gpurun_out/r06_default_bench.time
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
tail -3 gpurun_out/r06_default_bench.time
ESMDIFF_LIB=$PWD/esmdiff_amd/lib/libesmdiff_hip_fr2.so LAYERS=48 timeout 300 python scratch/r06_frames_race.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_frames_race.txt
ESMDIFF_LIB=$PWD/esmdiff_amd/lib/libesmdiff_hip_fr2.so LAYERS=1 timeout 300 python scratch/r06_frames_race.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_frames_race.txt
cat gpurun_out/r06_frames_race.txt
