#!/bin/bash
# like k8sweep.sh, on the colon workload (config 5): bash tools/k8sweep_colon.sh "name|ENV=1" ...
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  out=$(env $envs python bench.py --no-cpu-baseline --no-grad-steps --workload colon --res 1024 --spp 256 --grid 32 --fp16 --steps 15 --warmup 3 2>/dev/null | tail -1)
  python - "$name" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print(f"colon {sys.argv[1]:22s} {d['value']:8.2f} renders/s  step {d['ms_per_step']:.3f} ms  K8 {d['roofline']['avg_kernel_ms']:.3f} ms")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
