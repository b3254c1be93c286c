#!/bin/bash
# kernel resource usage of one .hip file: name, SGPRs, VGPRs, spills, scratch, LDS, occupancy
f=${1:-fireflies_amd/csrc/ffx_trace.hip}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-inline-asm "$@" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n,l; n=$3; l=""} /TotalSGPRs|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size/{l=l" | "$0} END{print n,l}' | sed 's/  */ /g' | c++filt | sed -E "s/\(ShadeK[^)]*\)/()/" | cut -c1-260
