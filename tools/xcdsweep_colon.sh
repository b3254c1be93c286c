#!/bin/bash
# config 5 (colon 1024x1024x256 fp16): renders/s and K8's FETCH_SIZE per launch over the blocked XCD interleave (FFX_XCD_REMAP = workgroups
# per XCD per 8*B block): bash tools/xcdsweep_colon.sh 128 256 512 ...
R=$(pwd); OUT=$R/gpurun_out/xcdsweep_colon; rm -rf $OUT; mkdir -p $OUT
A="--workload colon --res 1024 --spp 256 --grid 32 --fp16 --no-cpu-baseline --no-grad-steps"
export FFX_BENCH_PREFLIGHT=0
for m in "$@"; do
  export FFX_XCD_REMAP=$m
  v=$(cd $R && python bench.py $A --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['roofline']['avg_kernel_ms_after_bracket'],3))")
  (cd /tmp && export TMPDIR=/tmp FFX_BENCH_SETTLE=0 && cd $R && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/m${m} -o p --output-format csv -- python3 bench.py $A --steps 3 --warmup 1 > /dev/null 2>&1)
  f=$(python - "$OUT/m${m}" <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_fwd_pk" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
print(round(sum(vals) / max(len(vals), 1) / 1024, 1), "MB raw per launch, n", len(vals))
PY
)
  echo "FFX_XCD_REMAP=$m: renders/s, K8 ms: $v ; FETCH_SIZE $f"
  rm -rf $OUT/m${m}
done
