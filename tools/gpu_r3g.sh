#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $R/r3g_tests.log 2>&1; rc=$?
tail -3 $R/r3g_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile_round.sh r03a || exit 1
bash tools/profile_round.sh r03b5 --config 5 || exit 1
