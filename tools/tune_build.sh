#!/bin/bash
# tools/tune_build.sh name "-DRT_X=.. -DRT_Y=.." — build a tuning variant of librt_amd.so into tune/<name>.so
# (git-ignored; travels with gpurun).  Run a variant with: python bench.py --lib tune/<name>.so
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p ../../tune
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -w $2 -shared rt_gpu.hip rt_lbvh.hip rt_scene.cpp -o ../../tune/$1.so
