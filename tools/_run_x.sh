cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r4o_gputest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r4o_gputest.log
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],1), round(d.get("value_cold") or 0,1), round(d.get("grad_steps_per_sec") or 0,1), round(d.get("grad_steps_per_sec_nonlinear") or 0,1), d["roofline"]["kernel_alone_ms"])'
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$P" "100"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P" "20/5"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rfilter gaussian 2>/dev/null | python -c "$P" "gauss"
python tools/rftime.py vocalfold 2>&1 | grep -v amdgpu | tail -1
