#!/usr/bin/env python3
"""Collapse rocprofv3 counter_collection CSVs into a per-kernel table:
   python tools/pmc_table.py <dir> [kernel-substring]   ->  mean counter value per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")
                if sub in k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k[:60]] = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
