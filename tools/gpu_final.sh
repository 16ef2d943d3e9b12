#!/bin/bash
# end-of-round check on the GPU box: the whole GPU suite, smoke(), the default bench line (with the CPU baseline)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $R/final_tests.log 2>&1; rc=$?
tail -3 $R/final_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $R/final_smoke.log 2>&1 || { tail -5 $R/final_smoke.log; exit 1; }
tail -1 $R/final_smoke.log
timeout -k 10 600 python bench.py > $R/final_bench.json 2> $R/final_bench.err || { tail -5 $R/final_bench.err; exit 1; }
tail -1 $R/final_bench.json
