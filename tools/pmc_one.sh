#!/bin/bash
# counters of ONE kernel (name substring $1) over the launches of a command: tools/pmc_one.sh <kernel> <outdir> -- python3 tools/x.py ...
# one rocprofv3 --pmc pass per counter set (kernel-trace only beside it), mean per launch printed per counter
K=$1; OUT=$2; shift 3
R=$(pwd); mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_GDS SQ_INSTS_FLAT SQ_WAIT_ANY SQ_INST_CYCLES_VALU"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p --output-format csv -- "$@" > $OUT/p$i.log 2>&1)
done
cd $R
python3 - "$K" $OUT <<'PY'
import csv, glob, sys, collections
k, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if k in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"{c:28s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
PY
