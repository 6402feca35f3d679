#!/usr/bin/env python3
"""VALU-side summary of the render kernels from rocprofv3 --pmc passes and a kernel trace of the same
program.  usage: pmc_valu.py <pmc_dir> [<pmc_dir> ...] --trace <kernel_trace_dir> --out <json> [--note text]

Per kernel and per launch: wave-level VALU and SALU instructions, lane occupancy of the VALU
instructions (SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)), and the share of a SIMD's issue capacity
the kernel used for each kind.  The capacities are measured (tools/issue_rate_probe.hip, eight waves per
SIMD, nominal 2.4 GHz): one wave64 VALU instruction per 2.41 cycles and one SALU instruction per 4.34
cycles per SIMD; a mix costs about the larger of the two plus a tenth.  (SQ_ACTIVE_INST_VALU equals the
instruction count, so it says nothing about how long an instruction holds the SIMD.)
busy = instructions x cycles each / (launch seconds x clock x 1024 SIMDs).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

CLOCK_HZ = 2.4e9
SIMDS = 256 * 4
VALU_CYCLES = 2.41
SALU_CYCLES = 4.34


def short(k):
    for n in ("trace_kernel", "shade_kernel", "raygen_kernel", "accumulate_kernel", "tail_kernel"):
        if n in k:
            return n
    return None


ap = argparse.ArgumentParser()
ap.add_argument("pmc", nargs="+")
ap.add_argument("--trace", required=True)
ap.add_argument("--out", required=True)
ap.add_argument("--note", default="")
a = ap.parse_args()

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for d in a.pmc:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
dur, nl = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(a.trace, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            nl[k] += 1

out = {"note": a.note, "clock_hz": CLOCK_HZ, "simds": SIMDS, "valu_cycles_per_instruction": VALU_CYCLES,
       "salu_cycles_per_instruction": SALU_CYCLES, "kernels": {}}
for k in sorted(tot):
    per = {c: tot[k][c] / max(len(disp[k][c]), 1) for c in tot[k]}
    sec = dur[k] / max(nl[k], 1)
    e = {"launches_profiled": max(len(v) for v in disp[k].values()), "avg_launch_s": sec, "per_launch": per}
    if "SQ_INSTS_VALU" in per and "SQ_THREAD_CYCLES_VALU" in per and per["SQ_INSTS_VALU"] > 0:
        e["valu_lane_occupancy"] = per["SQ_THREAD_CYCLES_VALU"] / (64.0 * per["SQ_INSTS_VALU"])
    if "SQ_INSTS_VALU" in per and sec > 0:
        e["valu_issue_busy"] = VALU_CYCLES * per["SQ_INSTS_VALU"] / (sec * CLOCK_HZ * SIMDS)
    if "SQ_INSTS_SALU" in per and sec > 0:
        e["salu_issue_busy"] = SALU_CYCLES * per["SQ_INSTS_SALU"] / (sec * CLOCK_HZ * SIMDS)
    if "SQ_INSTS_VALU_FLOPS_FP32" in per and sec > 0:
        e["fp32_tflops"] = per["SQ_INSTS_VALU_FLOPS_FP32"] / sec * 1e-12
    out["kernels"][k] = e
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out, indent=1))
