cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r4l_default.json 2> gpurun_out/r4l_default.err; echo "rc $?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r4l_b20.json 2> gpurun_out/r4l_b20.err; echo "rc $?"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rfilter gaussian > gpurun_out/r4l_gauss.json 2> gpurun_out/r4l_gauss.err; echo "rc $?"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --material diffuse > gpurun_out/r4l_diffuse.json 2> gpurun_out/r4l_diffuse.err; echo "rc $?"
python - <<'PY'
import json
for f in ('r4l_default','r4l_b20','r4l_gauss','r4l_diffuse'):
    d=json.load(open(f'gpurun_out/{f}.json'))
    r=d['roofline']
    print(f, {k:round(d.get(k),1) for k in ('value','value_cold','value_overlapped','grad_steps_per_sec','grad_steps_per_sec_nonlinear')}, 'k8 in-loop', round(r.get('avg_kernel_ms'),4), 'alone', round(r.get('kernel_alone_ms'),4), 'traffic', r.get('traffic'), (r.get('valu_issue') or {}).get('frac'), (r.get('valu_issue') or {}).get('frac_operand_forms'), d.get('grad_step_paths'))
    if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline'])
PY
