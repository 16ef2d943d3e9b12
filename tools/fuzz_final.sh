#!/bin/bash
# end-of-round fuzz on the final build: round 2's RT_DEBUG matrix, a long run on the production paths and with the in-step walk
# forced, the refilling walk forced (its distance-only form and shared box test are new), the allocation-free entry point
bash tools/fuzz_matrix.sh || exit 1
RT_DEBUG=0 timeout -k 10 400 python -u tools/fuzz_gpu.py --first 300000 --count 40000 --seconds 300 > gpurun_out/fuzzF_prod.log 2>&1 || { echo "production FAILED: $(tail -2 gpurun_out/fuzzF_prod.log)"; exit 1; }
echo "production: $(tail -1 gpurun_out/fuzzF_prod.log)"
RT_DEBUG=256 timeout -k 10 300 python -u tools/fuzz_gpu.py --first 400000 --count 20000 --seconds 200 > gpurun_out/fuzzF_256.log 2>&1 || { echo "256 FAILED: $(tail -2 gpurun_out/fuzzF_256.log)"; exit 1; }
echo "in-step forced: $(tail -1 gpurun_out/fuzzF_256.log)"
RT_DEBUG=16 timeout -k 10 300 python -u tools/fuzz_gpu.py --first 500000 --count 20000 --seconds 200 > gpurun_out/fuzzF_16.log 2>&1 || { echo "16 FAILED: $(tail -2 gpurun_out/fuzzF_16.log)"; exit 1; }
echo "k_ao_walk forced: $(tail -1 gpurun_out/fuzzF_16.log)"
timeout -k 10 200 python -u tools/fuzz_noalloc.py > gpurun_out/fuzzF_noalloc.log 2>&1 || { echo "noalloc FAILED: $(tail -2 gpurun_out/fuzzF_noalloc.log)"; exit 1; }
echo "noalloc: $(tail -1 gpurun_out/fuzzF_noalloc.log)"
