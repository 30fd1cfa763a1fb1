#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06_t12.log
cp gpurun_out/parity_strict.json gpurun_out/r06_parity_strict.json 2>/dev/null
cp gpurun_out/parity_fullwidth.json gpurun_out/r06_parity_fullwidth.json 2>/dev/null
tail -12 gpurun_out/r06_t12.log
