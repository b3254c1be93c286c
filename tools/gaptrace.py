#!/usr/bin/env python3
"""gaptrace.py — between two renders of a gradient step: from a rocprofv3 kernel trace (csv), for every pair of consecutive fused render launches the
dispatches that start between the END of the first and the START of the second (the serial pattern side), with offsets from the render's end.

    rocprofv3 --kernel-trace -d gpurun_out/gt -o t --output-format csv -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-render-steps
    python tools/gaptrace.py gpurun_out/gt [anchor-substring]
"""
import csv
import glob
import os
import statistics
import sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_render_fwd_pk"
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:60]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    gaps, comp = [], {}
    for a, b in zip(idx[:-1], idx[1:]):
        end_a, start_b = rows[a][1], rows[b][0]
        if not 0 < start_b - end_a < 200_000:
            continue
        gaps.append((start_b - end_a) / 1e3)
        for r in rows[a + 1:b]:
            if r[0] >= end_a - 2000:  # (dispatches of the serial part; the side chain of the next pose overlaps the render and starts earlier)
                comp.setdefault(r[2], []).append(((r[0] - end_a) / 1e3, (r[1] - r[0]) / 1e3))
    print(f"{len(gaps)} render-to-render gaps: median {statistics.median(gaps):.1f} us, min {min(gaps):.1f}, max {max(gaps):.1f}")
    for k, v in sorted(comp.items(), key=lambda kv: statistics.median(x[0] for x in kv[1])):
        print(f"  {k:46s} in {len(v):3d} gaps: starts {statistics.median(x[0] for x in v):7.1f} us after the render's end, runs {statistics.median(x[1] for x in v):6.1f} us")


if __name__ == "__main__":
    main()
