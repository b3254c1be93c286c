#!/bin/bash
# builds fireflies_amd/csrc/_stats/libffx_hip_<name>.so with extra -D flags: tools/build_variant_lib.sh name -DFOO ...
set -e
name=$1; shift
cd "$(dirname "$0")/../fireflies_amd/csrc"
mkdir -p _stats/$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-inline-asm $*"
for f in ffx_splat.hip ffx_scene.hip ffx_trace.hip ffx_bins.hip ffx_bvh.cpp ffx_rng.cpp; do /opt/rocm/bin/hipcc $F -c $f -o _stats/$name/${f%.*}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _stats/libffx_hip_$name.so _stats/$name/*.o
echo built _stats/libffx_hip_$name.so
