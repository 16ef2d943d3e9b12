#!/bin/bash
# tools/tune_build.sh <name> "<flags for every TU>" ["<flags for rt_bvh.hip>" ["<flags for rt_gpu.hip>"]] — builds tune/<name>.so, a variant
# of librt_amd.so with other -D tunables / compiler flags, for `bench.py --lib tune/<name>.so` or
# tools/run_variants.sh.  The third / fourth argument replace the Makefile's BVH_FLAGS / GPU_FLAGS (scheduler strategies).
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../../tune
T=$(mktemp -d)
F="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -w"
B="${3--mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -slp-threshold=1}"
G="${4--mllvm -amdgpu-sched-strategy=max-ilp -mllvm -disable-machine-licm -fno-slp-vectorize}"
/opt/rocm/bin/hipcc $F $2 $B -c rt_bvh.hip -o $T/rt_bvh.o &
/opt/rocm/bin/hipcc $F $2 $G -c rt_gpu.hip -o $T/rt_gpu.o &
/opt/rocm/bin/hipcc $F $2 -c rt_lbvh.hip -o $T/rt_lbvh.o &
/opt/rocm/bin/hipcc $F $2 -c rt_scene.cpp -o $T/rt_scene.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/rt_gpu.o $T/rt_bvh.o $T/rt_lbvh.o $T/rt_scene.o -o ../../tune/$1.so
rm -rf $T
