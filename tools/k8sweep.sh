#!/bin/bash
# A/B sweep of K8 variants through the bench loop (randomise + refit + render, 60 steps): one line per variant.
# usage (on the GPU box): bash tools/k8sweep.sh "name|ENV=1 ENV2=x" ...   (FFX_LIB etc. go in the env part)
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  out=$(env $envs python bench.py --steps 60 --warmup 10 --no-grad-steps --no-cpu-baseline $K8SWEEP_ARGS 2>/dev/null | tail -1)
  python - "$name" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print(f"{sys.argv[1]:28s} {d['value']:8.1f} renders/s  step {d['ms_per_step']:.4f} ms  K8 {d['roofline']['avg_kernel_ms']:.4f} ms")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
