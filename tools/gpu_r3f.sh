#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $R/r3f_tests.log 2>&1; rc=$?
tail -3 $R/r3f_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline > $R/r3f_bench5.json 2> $R/r3f_bench5.err || exit 1
RT_DEBUG=64 timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline > $R/r3f_bench5g.json 2> $R/r3f_bench5g.err || exit 1
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/r3f_bench3.json 2> $R/r3f_bench3.err || exit 1
timeout -k 10 300 python bench.py --config 4 --width 1920 --height 1080 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline > $R/r3f_bench4.json 2> $R/r3f_bench4.err || exit 1
python - <<PY
import json
for f in ("r3f_bench5","r3f_bench5g","r3f_bench3","r3f_bench4"):
    d=json.load(open("$R/"+f+".json")); print(f, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["config"]["frame_checksum"])
PY
