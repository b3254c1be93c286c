#!/usr/bin/env python3
"""What the GPU does in ONE step, from a rocprofv3 kernel trace: the dispatches between consecutive launches of an anchor
kernel (default: the render kernel) in the steady state of a bracket — names, durations, and the idle gaps between them.
Answers "how many copyBuffer / fill / reduce launches does a step really issue" (tools/copyprobe.py only sees host calls).

    rocprofv3 --kernel-trace -d gpurun_out/trace -o t --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
    python tools/steptrace.py gpurun_out/trace [anchor-substring] > profiles/<tag>_steptrace.txt
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("void ", "")
    for cut in ("(", "<"):
        if cut in name and not name.startswith("at::"):
            name = name.split(cut)[0]
    return name[:60]


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_render_fwd_pk"
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    print(f"{len(rows)} dispatches, {len(idx)} of '{anchor}'")
    # brackets: runs of anchors whose spacing is regular; report per-step composition for each run of >= 6 anchors
    steps = []
    for a, b in zip(idx[:-1], idx[1:]):
        seg = rows[a:b]  # the anchor launch and everything dispatched until the next one
        period = rows[b][0] - rows[a][0]
        steps.append((period, seg))
    if not steps:
        return
    periods = sorted(p for p, _ in steps)
    med = periods[len(periods) // 2]
    steady = [(p, seg) for p, seg in steps if p < 1.5 * med]
    print(f"median period {med / 1e3:.1f} us; {len(steady)} steady steps")
    # (a gradient step: one with an adjoint launch or — the fused forward + adjoint, ffx_render_fwd_adjoint — a pattern launch)
    is_grad = lambda seg: any("render_bwd" in r[2] or "k_pattern_bwd" in r[2] for r in seg)  # noqa: E731
    for label, pick in (("steps without an adjoint / pattern launch (render bracket)", lambda seg: not is_grad(seg)),
                        ("steps with an adjoint / pattern launch (gradient bracket)", is_grad)):
        sel = [(p, seg) for p, seg in steady if pick(seg)]
        if not sel:
            continue
        n = len(sel)
        cnt, dur = collections.Counter(), collections.Counter()
        busy = 0
        for p, seg in sel:
            for s, e, name in seg:
                cnt[short(name)] += 1
                dur[short(name)] += e - s
            # union of busy intervals (streams overlap)
            iv = sorted((s, e) for s, e, _ in seg)
            cur_s, cur_e = iv[0]
            for s, e in iv[1:]:
                if s > cur_e:
                    busy += cur_e - cur_s
                    cur_s, cur_e = s, e
                else:
                    cur_e = max(cur_e, e)
            busy += cur_e - cur_s
        per = sum(p for p, _ in sel) / n
        print(f"\n== {label}: {n} steps, period {per / 1e3:.1f} us, GPU busy (any stream) {busy / n / 1e3:.1f} us, idle {(per - busy / n) / 1e3:.1f} us per step")
        print(f"{'kernel':62s} {'per step':>8s} {'avg us':>8s} {'us/step':>8s}")
        for name, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
            print(f"{name:62s} {c / n:8.2f} {dur[name] / c / 1e3:8.1f} {dur[name] / n / 1e3:8.1f}")


if __name__ == "__main__":
    main()
