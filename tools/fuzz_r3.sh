#!/bin/bash
# round 3 fuzz campaign: random scenes against the oracle on the production paths and with the in-step gate walk forced on
# everywhere (RT_DEBUG 256), also with the capped work queue (264) and the inline shadow kernel (260)
for d in 0 256 264 260; do
  RT_DEBUG=$d timeout -k 10 200 python -u tools/fuzz_gpu.py --first $((120000 + d * 10)) --count 1500 --seconds 150 > gpurun_out/fuzz3_dbg$d.log 2>&1 || { echo "RT_DEBUG=$d FAILED: $(tail -2 gpurun_out/fuzz3_dbg$d.log)"; exit 1; }
  echo "RT_DEBUG=$d: $(tail -1 gpurun_out/fuzz3_dbg$d.log)"
done
RT_DEBUG=256 timeout -k 10 300 python -u tools/fuzz_gpu.py --first 190000 --count 800 --seconds 200 --scale 3 > gpurun_out/fuzz3_dbg256s3.log 2>&1 || { echo "scale 3 FAILED"; exit 1; }
echo "RT_DEBUG=256 scale 3: $(tail -1 gpurun_out/fuzz3_dbg256s3.log)"
