cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "adjoint or cached or cache or gaussian or overflow or sparse" 2>&1 | tail -2
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],1), round(d.get("grad_steps_per_sec") or 0,1), round(d.get("grad_steps_per_sec_nonlinear") or 0,1), d["grad_kernels_ms"])'
for i in 1 2; do python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P" "100"; done
