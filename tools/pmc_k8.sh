#!/bin/bash
# PMC passes over K8 alone (tools/k8once.py), wide vs binary walk: bash tools/pmc_k8.sh <outdir>
R=$(pwd); OUT=$R/gpurun_out/${1:-pmc_k8}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in 1 0; do
 export FFX_WIDE=$w
 i=0
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
            "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES SQ_INST_CYCLES_SMEM" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --kernel-trace -d $OUT/w${w}_$i -o p --output-format csv -- python3 tools/k8once.py 3 > /dev/null 2>&1)
 done
 mkdir -p $OUT/all_w$w; cp -r $OUT/w${w}_* $OUT/all_w$w/ 2>/dev/null
 (cd $R && python tools/pmc_table.py $OUT/all_w$w k_render_fwd_pk > $OUT/k8_w$w.json)
 rm -rf $OUT/w${w}_* $OUT/all_w$w
done
cat $OUT/k8_w1.json $OUT/k8_w0.json | grep -E "mean|k_render" | sed 's/"launches": 3//' | tr -s " " | paste - - - - | head -80
