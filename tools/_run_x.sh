cd $GRAFT_REPO_ROOT
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],1), round(d.get("grad_steps_per_sec") or 0,1), round(d.get("grad_steps_per_sec_nonlinear") or 0,1), d["roofline"]["kernel_alone_ms"])'
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rfilter gaussian 2>/dev/null | python -c "$P" "gauss"
python tools/rftime.py vocalfold 2>&1 | grep -v amdgpu | tail -1
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gaussian" 2>&1 | tail -2
