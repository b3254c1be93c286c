#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline fields on an MI355X box (run via gpurun
# from the repo root):  bash tools/collect_profiles.sh <tag> [bench.py arguments]
# <tag> = r<N> for bench.py's default workload, r<N>colon with `--workload colon --res 1024 --spp 256 --grid 32 --fp16`
# (bench.py's profile_key finds the files by that name).
# Every rocprofv3 pass runs the program directly after `--`; PMC passes are separate from the
# kernel-trace/stats pass and from each other (FETCH_SIZE, WRITE_SIZE, SQ instruction mix).
set -u
TAG=${1:-r3}
shift
X="$*"   # extra bench.py arguments: the workload
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the profiled runs time and count the bracket's own launches only: no clock-settle launches (the same kernel on ONE pose, cheaper than
# the average random pose: they would pull the per-kernel averages down) and no preflight renders (bench.py: FFX_BENCH_SETTLE / _PREFLIGHT)
# ... and no gaussian-film brackets (round 5: bench.py's default line carries them; here they get a pass of their own, r<N>gauss_*)
export FFX_BENCH_SETTLE=0 FFX_BENCH_PREFLIGHT=0 FFX_BENCH_EXTRA_BRACKETS=0 FFX_BENCH_GAUSSIAN=0
B="--steps 20 --warmup 3 --no-cpu-baseline $X"
(cd $R && rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py $B > $OUT/bench_under_rocprof.json 2> $OUT/stats.log)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-grad-steps $X > /dev/null 2>&1)
  (cd $R && rocprofv3 --pmc $c --kernel-trace -d $OUT/gpmc_$c -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-render-steps $X > /dev/null 2>&1)
done
if [ -z "$X" ]; then  # the gaussian film's kernels (k_render_fwd_pk<..., RF[, RFC]>, k_rf_gather, k_render_bwd_cached_filtered): kernel stats + the line
  (cd $R && rocprofv3 --kernel-trace --stats -d $OUT/gstats -o s --output-format csv -- python3 bench.py $B --rfilter gaussian > $OUT/gauss_bench_under_rocprof.json 2> $OUT/gstats.log)
  G=$(find $OUT/gstats -name "*kernel_stats.csv" | head -1)
  [ -n "$G" ] && cp "$G" $R/profiles/${TAG}gauss_kernel_stats.csv
  tail -1 $OUT/gauss_bench_under_rocprof.json > $R/profiles/${TAG}gauss_bench_under_rocprof.json
fi
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_INST_CYCLES_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --kernel-trace -d $OUT/sq_$i -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-grad-steps $X > /dev/null 2>&1)
done
cd $R
python tools/summarize_profile.py $TAG $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/summary.txt 2>&1
python tools/summarize_profile.py ${TAG}grad $OUT/stats $OUT/gpmc_FETCH_SIZE $OUT/gpmc_WRITE_SIZE >> $OUT/summary.txt 2>&1
rm -f profiles/${TAG}grad_kernel_stats.csv
mkdir -p $OUT/sq_all; cp -r $OUT/sq_[0-9] $OUT/sq_all/
python tools/pmc_table.py $OUT/sq_all k_render_fwd_pk > profiles/${TAG}_sq_instruction_mix.json
tail -1 $OUT/bench_under_rocprof.json > profiles/${TAG}_bench_under_rocprof.json
if [ -z "$X" ]; then  # workload-independent pieces: with the default workload only
python tools/microbench.py > profiles/${TAG}_microbench.json 2> $OUT/microbench.log
# issue-rate microbenchmark (instruction counts are fixed by the inline-asm bodies; `grep -c` on the .s confirms)
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rates issue_rates.hip 2> $OUT/issue_rates_build.log && /tmp/issue_rates > $R/profiles/${TAG}_issue_rates.txt 2>&1)
fi
if [ -z "$X" ]; then
# the line as the driver runs it (no profiler, every informational bracket: gaussian film, low spp, configs[3] and [4]): what BENCH_r<N>.json will look like
(unset FFX_BENCH_SETTLE FFX_BENCH_PREFLIGHT FFX_BENCH_EXTRA_BRACKETS FFX_BENCH_GAUSSIAN; python bench.py --steps 20 --warmup 5 2> $OUT/bench_line.log | tail -1 > $OUT/bench_line.json) \
  && [ -s $OUT/bench_line.json ] && mv $OUT/bench_line.json profiles/${TAG}_bench_line.json
fi
# where K8's wave time goes (shader-clock stamps; needs the -DFFX_TIMERS build of the CURRENT sources: tools/build_variant_lib.sh timers -DFFX_TIMERS,
# before gpurun — a stale variant library fails to load (missing symbols) and used to leave an empty file behind)
FAILED=""
TL=fireflies_amd/csrc/_stats/libffx_hip_timers.so
if [ -f $TL ]; then  # (no mtime comparison: the snapshot that carries the tree to the GPU box does not keep the files' order in time; a stale variant fails to load)
  if [ -z "$X" ]; then FFX_LIB=$TL python tools/phaseclk.py vocalfold 64 8 > $OUT/phaseclk.txt 2> $OUT/phaseclk.log
  else FFX_LIB=$TL python tools/phaseclk.py colon 256 2 > $OUT/phaseclk.txt 2> $OUT/phaseclk.log; fi
  if [ $? -eq 0 ] && [ -s $OUT/phaseclk.txt ]; then mv $OUT/phaseclk.txt profiles/${TAG}_phaseclk.txt
  else FAILED="$FAILED phaseclk(see $OUT/phaseclk.log)"; rm -f profiles/${TAG}_phaseclk.txt; tail -3 $OUT/phaseclk.log; fi
else
  FAILED="$FAILED phaseclk(no current timers build: tools/build_variant_lib.sh timers -DFFX_TIMERS)"; rm -f profiles/${TAG}_phaseclk.txt
fi
if [ -n "${COLLECT_CPU_TABLE:-}" ]; then
  python tools/cpu_oracle_table.py > $OUT/cpu_oracle_table.json 2> $OUT/cpu_table.log && [ -s $OUT/cpu_oracle_table.json ] && mv $OUT/cpu_oracle_table.json profiles/${TAG}_cpu_oracle_table.json \
    || FAILED="$FAILED cpu_oracle_table"
fi
if [ -z "$X" ]; then
python tools/isa_mix.py > $OUT/isa_operand_forms.json 2> $OUT/isa_mix.log && mv $OUT/isa_operand_forms.json profiles/${TAG}_isa_operand_forms.json   # (static: needs no GPU, kept with the rest)
python tools/k8ab.py > profiles/${TAG}_k8ab.txt 2> $OUT/k8ab.log
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace.log 2>&1)
python tools/steptrace.py $OUT/trace > profiles/${TAG}_steptrace.txt 2>> $OUT/trace.log
rm -rf $OUT/trace
fi
# nothing empty leaves this script: a summary that came out empty (a pass that failed behind a redirect) is removed and reported, and the
# script exits non-zero — profiles/r3_phaseclk.txt was committed as a 0-byte file once
for f in profiles/${TAG}_* profiles/${TAG}grad_*; do
  [ -e "$f" ] || continue
  if [ ! -s "$f" ]; then FAILED="$FAILED $(basename $f)(empty)"; rm -f "$f"; fi
done
for need in kernel_stats.csv pmc_summary.json sq_instruction_mix.json bench_under_rocprof.json; do
  [ -s profiles/${TAG}_$need ] || FAILED="$FAILED ${TAG}_$need(missing)"
done
[ -s profiles/${TAG}grad_pmc_summary.json ] || FAILED="$FAILED ${TAG}grad_pmc_summary.json(missing)"
python - "$TAG" <<'PYEOF' || FAILED="$FAILED kernel-names"
import json, sys
tag = sys.argv[1]
g = json.load(open(f"profiles/{tag}grad_pmc_summary.json"))
want = ["k_render_fwd_pk<", "k_pattern_step<5>", "k_render_bwd_cached"]  # (round 6: the pattern side of a step is ONE launch)
missing = [w for w in want if not any(k.startswith(w) for k in g)]
def adjoint_instance(k):  # k_render_fwd_pk<R, WIDE, MATM, ADJ, RF>
    import re
    m = re.search(r"k_render_fwd_pk<([^>]*)>", k)
    a = [t.strip() for t in m.group(1).split(",")] if m else []
    return len(a) > 3 and a[3] == "true"
if not any(adjoint_instance(k) for k in g):
    missing.append("k_render_fwd_pk<..., true> (forward + adjoint)")
if missing:
    print("collect_profiles: the gradient bracket's PMC summary lacks", missing, file=sys.stderr)
    sys.exit(1)
PYEOF
# ship the small summaries back (profiles/ is not merged by gpurun, gpurun_out/ is)
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}* gpurun_out/profiles_$TAG/
rm -rf $OUT/stats $OUT/pmc_* $OUT/gpmc_* $OUT/sq_*
tail -5 $OUT/summary.txt
if [ -n "$FAILED" ]; then echo "collect_profiles.sh: FAILED:$FAILED" >&2; exit 1; fi
