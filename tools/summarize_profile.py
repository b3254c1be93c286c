#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into small, committed summaries under profiles/.

    python tools/summarize_profile.py r1 gpurun_out/prof_r1 [gpurun_out/pmc_fetch gpurun_out/pmc_write ...]

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table, ffx kernels + top torch
kernels) and profiles/<tag>_pmc_summary.json (per-kernel averages of every counter found, plus the
HBM traffic estimate for the render kernels with the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 64 B per 128-B request for wide reads, so it is doubled;
both raw and corrected numbers are kept because narrow / scalar accesses are uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    os.makedirs("profiles", exist_ok=True)
    f = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if r["Name"].startswith(("k_", "void k_"))] + [r for r in rows if not r["Name"].startswith(("k_", "void k_"))][:8]
    with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as out:
        w = csv.DictWriter(out, fieldnames=rows[0].keys())
        w.writeheader()
        for r in keep:
            r = dict(r)
            r["Name"] = r["Name"][:120]
            w.writerow(r)
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if name.startswith("k_"):
                    pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary = {}
    for k, cs in pmc.items():
        summary[k] = {c: {"avg": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            fe, wr = summary[k]["FETCH_SIZE"]["avg"], summary[k]["WRITE_SIZE"]["avg"]
            summary[k]["hbm_bytes_raw"] = (fe + wr) * 1024
            summary[k]["hbm_bytes_corrected"] = (2 * fe + wr) * 1024
    json.dump(summary, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
    print(open(f"profiles/{tag}_kernel_stats.csv").read()[:1500])
    print(json.dumps({k: {c: v for c, v in s.items() if not isinstance(v, dict)} for k, s in summary.items()}, indent=1))


if __name__ == "__main__":
    main()
