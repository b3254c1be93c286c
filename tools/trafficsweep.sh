#!/bin/bash
# K8's speed (bench loop, 60 steps) and HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over tools/k8once.py) per launch
# configuration: bash tools/trafficsweep.sh "name|ENV=1 ENV2=x" ...   -> one line per configuration (corrected = 2 x FETCH + WRITE, KB -> MB)
# TRAFFIC_CMD = the program the counter passes run (default: four renders of the vocal fold); K8SWEEP_ARGS = the speed leg's bench arguments
R=$(pwd); OUT=$R/gpurun_out/trafficsweep; rm -rf $OUT; mkdir -p $OUT
export FFX_BENCH_GAUSSIAN=0 FFX_BENCH_EXTRA_BRACKETS=0
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  line=$(cd $R && env $envs bash tools/k8sweep.sh "$name|$envs")
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && cd $R && env $envs rocprofv3 --pmc $c --kernel-trace -d $OUT/${name}_$c -o p --output-format csv -- ${TRAFFIC_CMD:-python3 tools/k8once.py 4} > /dev/null 2>&1)
  done
  python - "$OUT/${name}" "$line" <<'PY'
import csv, glob, sys
def mean(c):
    v = []
    for f in glob.glob(sys.argv[1] + "_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_render_fwd_pk" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c:
                v.append(float(r["Counter_Value"]))
    return sum(v) / max(len(v), 1)
fe, wr = mean("FETCH_SIZE"), mean("WRITE_SIZE")
print(f"{sys.argv[2]}   FETCH {fe / 1e3:.1f} MB raw, WRITE {wr / 1e3:.1f} MB, corrected {(2 * fe + wr) * 1024 / 1e6:.1f} MB")
PY
done
