# like run_variants.sh for config 4 at half its stated size (--steps 2)
for v in "$@"; do timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --config 4 --lib tune/$v.so > gpurun_out/var4_$v.json 2>gpurun_out/var4_$v.err || exit 1; python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print(sys.argv[1], d['value'], d['ms_per_step'], {a:k[a] for a in ('trace','trace_bvh','shadow','ao','ao_bvh')}, d['config']['frame_checksum'])" gpurun_out/var4_$v.json; done
