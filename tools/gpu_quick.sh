#!/bin/bash
# quick check: the parity tests of the traversal paths, then the config-3 / config-5 bench lines (no CPU baseline) with k_bvh's lane diagnostics
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
T=${1:-q}
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $R/${T}_tests.log 2>&1; rc=$?
tail -3 $R/${T}_tests.log
[ $rc -eq 0 ] || exit $rc
RT_BVH_DIAG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $R/${T}_c3.json 2> $R/${T}_c3.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --config 5 > $R/${T}_c5.json 2> $R/${T}_c5.err || exit 1
grep "k_bvh" $R/${T}_c3.err
python - $T <<'PY'
import json, sys
T = sys.argv[1]
for f in (f"{T}_c3.json", f"{T}_c5.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["config"]["frame_checksum"])
PY
