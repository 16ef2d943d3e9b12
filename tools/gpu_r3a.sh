#!/bin/bash
# round 3, first GPU call: full GPU test suite, bench (config 3), the two measurement programs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $R/r3a_tests.log 2>&1; rc=$?
tail -5 $R/r3a_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > $R/r3a_bench.json 2> $R/r3a_bench.err || exit 1
cat $R/r3a_bench.json
timeout -k 10 120 ./tools/micro/valu_rate > $R/r3a_valu_rate.jsonl 2>&1 || exit 1
rm -rf $R/r3a_calib
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/r3a_calib -- ./tools/micro/fetch_calib > $R/r3a_calib.json 2> $R/r3a_calib.err || exit 1
cat $R/r3a_calib.json
