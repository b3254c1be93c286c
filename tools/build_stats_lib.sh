#!/bin/bash
# builds fireflies_amd/csrc/_stats/libffx_hip_stats.so: the product library with -DFFX_STATS (walk counters)
set -e
cd "$(dirname "$0")/../fireflies_amd/csrc"
mkdir -p _stats
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-inline-asm -DFFX_STATS"
rm -f _stats/*.o
for f in ffx_splat.hip ffx_scene.hip ffx_trace.hip ffx_bins.hip ffx_bvh.cpp ffx_rng.cpp; do /opt/rocm/bin/hipcc $F -c $f -o _stats/${f%.*}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _stats/libffx_hip_stats.so _stats/*.o
echo built _stats/libffx_hip_stats.so
