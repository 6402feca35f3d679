#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter.
usage: pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = "trace" if "trace_kernel" in k else "shade" if "shade_kernel" in k else "raygen" if "raygen" in k else \
                "accumulate" if "accumulate" in k else k[:40]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        n = len(calls[k][c])
        print("   %-28s total %.6g   per dispatch %.6g   (%d dispatches)" % (c, tot[k][c], tot[k][c] / max(n, 1), n))
