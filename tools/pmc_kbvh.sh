#!/bin/bash
# tools/pmc_kbvh.sh <tag> — one SQ counter pass over the bench (1 step) on the GPU box; prints per-kernel sums.
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
rm -rf $R/${TAG}_pmc_sq
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/${TAG}_pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/${TAG}_pmc_sq.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/${TAG}_pmc_sq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0][:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (row["Dispatch_Id"], k)
    if key not in seen: seen.add(key); n[k] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:8]:
    print(k, "launches", n[k], {a: round(b / 1e6, 1) for a, b in c.items()}, "(millions, summed over launches)")
PY
