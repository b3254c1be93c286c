#!/bin/bash
# round 6 A/B through the bench loop: tools/r6ab.sh OUT "name|ENV=.. ENV=..|bench args" ...   (one JSON line per variant, condensed)
out=$1; shift
mkdir -p $out
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; args=${rest#*|}
  env $envs python bench.py --no-cpu-baseline $args > $out/$name.json 2> $out/$name.err || echo "$name FAILED"
  python - "$out/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    keys = ["value", "grad_steps_per_sec", "grad_steps_per_sec_nonlinear", "value_gaussian", "value_16spp", "value_10spp_gaussian"]
    print(sys.argv[2], " ".join(f"{k}={d[k]:.0f}" for k in keys if isinstance(d.get(k), (int, float))), f"k8_alone={d['roofline'].get('kernel_alone_ms', 0):.4f}", flush=True)
except Exception as e:
    print(sys.argv[2], "no line:", e, flush=True)
PY
done
