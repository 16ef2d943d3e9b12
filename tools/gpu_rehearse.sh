#!/bin/bash
# N-rank rehearsal of bench.py on ONE GPU: gloo backend, all ranks share the card (slabs staged through the host).
# Checks the tile split / gather / unpack path at world sizes 2, 4, 8 with GPU rendering: frame_checksum must equal N = 1's.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 > $R/reh_1.json 2> $R/reh_1.err || exit 1
for n in 2 4 8; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 2 --warmup 1 --backend gloo --no-cpu-baseline > $R/reh_$n.json 2> $R/reh_$n.err || { tail -5 $R/reh_$n.err; exit 1; }
done
python - <<'PY'
import json
for n in (1, 2, 4, 8):
    d = json.loads([l for l in open(f"gpurun_out/reh_{n}.json") if l.startswith("{")][-1])
    print(n, d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["frame_checksum"], d["config"]["rays_per_frame"])
PY
