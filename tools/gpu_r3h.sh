#!/bin/bash
# round 3: parity tests + the reference's own scenes (configs 1, 2, 4) with the in-step gate walk (k_ao<WAVE>) on and off
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $R/r3h_tests.log 2>&1; rc=$?
tail -3 $R/r3h_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/time_configs.py --scale 2 > $R/r3h_cfg_wave.jsonl 2> $R/r3h_cfg_wave.err || exit 1
RT_DEBUG=512 timeout -k 10 200 python tools/time_configs.py --scale 2 > $R/r3h_cfg_lane.jsonl 2> $R/r3h_cfg_lane.err || exit 1
RT_DEBUG=256 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $R/r3h_c3_wave.json 2> $R/r3h_c3_wave.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > $R/r3h_c3_lane.json 2> $R/r3h_c3_lane.err || exit 1
python - <<'PY'
import json
for f in ("r3h_cfg_wave.jsonl","r3h_cfg_lane.jsonl"):
    for l in open("gpurun_out/"+f):
        d=json.loads(l); print(f, d["config"], d["frame_ms"], d["mrays_per_s"], d["kernel_ms"]["ao"], d["kernel_ms"]["ao_bvh"])
for f in ("r3h_c3_wave.json","r3h_c3_lane.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["config"]["frame_checksum"])
PY
