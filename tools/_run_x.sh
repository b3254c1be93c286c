cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],1), round(d.get("grad_steps_per_sec") or 0,1), round(d.get("grad_steps_per_sec_nonlinear") or 0,1), d["roofline"]["kernel_alone_ms"])'
for i in 1 2; do python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P" "principled 100"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" "principled 20/5"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --material diffuse 2>/dev/null | python -c "$P" "diffuse 100"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --material diffuse 2>/dev/null | python -c "$P" "diffuse 20/5"
timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "k8 or bins or hello or variant or k7 or lambert or albedo" 2>&1 | tail -2
