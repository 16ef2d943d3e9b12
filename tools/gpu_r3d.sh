#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "walk or config5 or overflow or other_configs" > $R/r3d_tests.log 2>&1; rc=$?
tail -3 $R/r3d_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 1 > $R/r3d_bench5.json 2> $R/r3d_bench5.err || exit 1
python - <<PY
import json
for f in ("r3d_bench5",):
    d=json.load(open("$R/"+f+".json")); print(f, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["config"]["frame_checksum"])
PY
