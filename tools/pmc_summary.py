#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch for kernels matching a pattern."""
import csv
import glob
import sys
from collections import defaultdict

pat = sys.argv[1] if len(sys.argv) > 1 else "score_polar"
files = sys.argv[2:] or glob.glob("gpurun_out/pmc*/**/*counter_collection.csv", recursive=True)
for f in sorted(files):
    acc = defaultdict(list)
    per_dispatch = defaultdict(lambda: defaultdict(float))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat not in row["Kernel_Name"]:
                continue
            per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
    for d in per_dispatch.values():
        for k, v in d.items():
            acc[k].append(v)
    print(f)
    for k, v in sorted(acc.items()):
        print(f"   {k:40s} n={len(v)} mean={sum(v)/len(v):.6g}")
