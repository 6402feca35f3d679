#!/usr/bin/env python3
"""Per-launch durations of one frame from a rocprofv3 --kernel-trace directory.  usage: level_times.py <dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "raygen" in r["Kernel_Name"]]
out = []
for r in rows[idx[-2]:idx[-1]]:
    n = r["Kernel_Name"]
    name = next((k for k in ("trace_kernel", "shade_kernel", "tail_kernel", "raygen", "accumulate", "tally", "resolve", "sample_keys") if k in n), n[:20])
    out.append("%s %.1f" % (name.replace("_kernel", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000))
print("  ".join(out))
