#!/bin/bash
# tools/profile_round.sh <tag> [bench.py arguments, e.g. --config 5] — run on the GPU box (through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>_stats/
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | hit/miss | SQ)  -> gpurun_out/<tag>_pmc*/
# Copy the summaries into profiles/ afterwards with tools/summarize_profile.py.
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf $R/${TAG}_stats $R/${TAG}_pmc_fetch $R/${TAG}_pmc_write $R/${TAG}_pmc_l2 $R/${TAG}_pmc_sq $R/${TAG}_pmc_sq2 $R/${TAG}_pmc_ta
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/${TAG}_stats -- python3 bench.py --no-cpu-baseline $EXTRA > $R/${TAG}_stats.log 2>&1 || exit 1
PM="--steps 1 --warmup 1 --no-cpu-baseline $EXTRA"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/${TAG}_pmc_fetch -- python3 bench.py $PM > $R/${TAG}_pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/${TAG}_pmc_write -- python3 bench.py $PM > $R/${TAG}_pmc_write.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/${TAG}_pmc_l2 -- python3 bench.py $PM > $R/${TAG}_pmc_l2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/${TAG}_pmc_sq -- python3 bench.py $PM > $R/${TAG}_pmc_sq.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/${TAG}_pmc_sq2 -- python3 bench.py $PM > $R/${TAG}_pmc_sq2.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $R/${TAG}_pmc_ta -- python3 bench.py $PM > $R/${TAG}_pmc_ta.log 2>&1 || exit 1
tail -1 $R/${TAG}_stats.log
