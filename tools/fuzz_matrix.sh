for d in 8 12 28 16 4; do
  RT_DEBUG=$d timeout -k 10 200 python -u tools/fuzz_gpu.py --first $((50000 + d * 1000)) --count 1200 --seconds 150 > gpurun_out/fuzz_dbg$d.log 2>&1
  echo "RT_DEBUG=$d: $(tail -1 gpurun_out/fuzz_dbg$d.log)"
done
RT_DEBUG=24 timeout -k 10 300 python -u tools/fuzz_gpu.py --first 90000 --count 800 --seconds 250 --scale 3 > gpurun_out/fuzz_dbg24s3.log 2>&1
echo "RT_DEBUG=24 scale 3: $(tail -1 gpurun_out/fuzz_dbg24s3.log)"
