#!/bin/bash
# K8 speed and HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) per XCD mapping: bash tools/xcdsweep.sh "0" "4" "32" ...
R=$(pwd); OUT=$R/gpurun_out/xcdsweep; rm -rf $OUT; mkdir -p $OUT
for m in "$@"; do
  export FFX_XCD_REMAP=$m
  (cd $R && bash tools/k8sweep.sh "xcd$m|FFX_XCD_REMAP=$m")
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && cd $R && rocprofv3 --pmc $c --kernel-trace -d $OUT/m${m}_$c -o p --output-format csv -- python3 tools/k8once.py 3 > /dev/null 2>&1)
    python - "$OUT/m${m}_$c" "$c" <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_fwd_pk" in r.get("Kernel_Name", "") and r.get("Counter_Name") == sys.argv[2]:
            vals.append(float(r["Counter_Value"]))
print("   ", sys.argv[2], "KB per launch:", round(sum(vals) / max(len(vals), 1), 1), "n", len(vals))
PY
  done
done
