#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $R/r3e_tests.log 2>&1; rc=$?
tail -5 $R/r3e_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline > $R/r3e_bench5.json 2> $R/r3e_bench5.err || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/r3e_bench3.json 2> $R/r3e_bench3.err || exit 1
python - <<PY
import json
for f in ("r3e_bench5","r3e_bench3"):
    d=json.load(open("$R/"+f+".json")); print(f, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["config"]["frame_checksum"])
PY
