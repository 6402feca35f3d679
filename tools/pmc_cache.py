#!/usr/bin/env python3
"""Cache-level summary of the render kernels from rocprofv3 --pmc passes (one pass per hardware block) and a
kernel trace of the same program.  usage: pmc_cache.py <pmc_dir> [<pmc_dir> ...] --trace <kernel_trace_dir> --out <json> [--note text]

Per kernel and per launch:
  L2 (TCC, summed over the XCDs' channels): requests, hits, misses, hit rate = HIT / (HIT + MISS), and the
     request rate against the measured ceiling for random 64-byte fetches (130 G requests/s, see below).  A TCC
     request moves at most one 128-byte line; the byte figure is requests x 128 B and therefore an upper bound
     (a 64-byte node fetch is one request as well).
  L1 (TCP, summed over the CUs): cache accesses (tag look-ups: one per 64-byte-wide piece of a wave's load, so a
     16-byte-per-lane load whose 64 lanes point at 64 different nodes costs 64 of them), the read and write
     requests L1 sent on to L2, L1's hit rate = 1 - TCC read requests / reads, and the access rate against the
     measured ceiling (820 G accesses/s).
  HBM (FETCH_SIZE / WRITE_SIZE, KiB; the read side doubled as gfx950 needs for wide coalesced reads).
Counters absent from the passes are left out of the output.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

L2_PEAK_GBS = 34500.0
# Ceilings of the cache hierarchy for RANDOM 64-byte fetches (what a tree walk does), measured with
# tools/node_fetch_probe.hip on the same chip (profiles/r02_node_fetch_probe.txt): chip-wide, the vector L1s sustain
# 820 G tag accesses/s and the L2s 130 G requests/s (= 8.3 TB/s of 64-byte nodes), whether a node is fetched by one lane
# as four 16-byte loads or by four lanes as one load each.
L1_PEAK_ACCESSES_PER_US = 820000.0
L2_PEAK_REQUESTS_PER_US = 130000.0
HBM_PEAK_GBS = 8000.0


def short(k):
    for n in ("trace_kernel", "shade_kernel", "raygen_kernel", "accumulate_kernel", "tail_kernel"):
        if n in k:
            return n
    return None


def main():
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
    out = {"note": a.note, "l1_peak_accesses_per_us": L1_PEAK_ACCESSES_PER_US, "l2_peak_requests_per_us": L2_PEAK_REQUESTS_PER_US, "l2_peak_GBs": L2_PEAK_GBS, "hbm_peak_GBs": HBM_PEAK_GBS, "kernels": {}}
    for k in sorted(tot):
        per = {c: tot[k][c] / max(len(disp[k][c]), 1) for c in tot[k]}
        sec = dur[k] / max(nl[k], 1)
        e = {"launches_profiled": max(len(v) for v in disp[k].values()), "avg_launch_s": sec, "per_launch": per}
        hit, miss = per.get("TCC_HIT_sum"), per.get("TCC_MISS_sum")
        if hit is not None and miss is not None and hit + miss > 0:
            e["l2_hit_rate"] = hit / (hit + miss)
        if "TCC_REQ_sum" in per and sec > 0:
            e["l2_request_bytes_upper_bound"] = per["TCC_REQ_sum"] * 128.0
            e["l2_GBs_upper_bound"] = per["TCC_REQ_sum"] * 128.0 / sec * 1e-9
            e["l2_frac_of_peak_upper_bound"] = e["l2_GBs_upper_bound"] / L2_PEAK_GBS
            e["l2_requests_per_us"] = per["TCC_REQ_sum"] / sec * 1e-6
            e["l2_request_frac_of_peak"] = e["l2_requests_per_us"] / L2_PEAK_REQUESTS_PER_US
        acc, rd = per.get("TCP_TOTAL_CACHE_ACCESSES_sum"), per.get("TCP_TCC_READ_REQ_sum")
        tr = per.get("TCP_TOTAL_READ_sum")
        if tr and rd is not None:
            e["l1_read_hit_rate"] = max(0.0, 1.0 - rd / tr)
        if acc is not None and sec > 0:
            e["l1_accesses_per_us"] = acc / sec * 1e-6
            e["l1_frac_of_peak"] = e["l1_accesses_per_us"] / L1_PEAK_ACCESSES_PER_US
        f_kib, w_kib = per.get("FETCH_SIZE"), per.get("WRITE_SIZE")
        if f_kib is not None or w_kib is not None:
            hbm = (f_kib or 0.0) * 1024 * 2.0 + (w_kib or 0.0) * 1024
            e["hbm_bytes_per_launch"] = hbm
            if sec > 0:
                e["hbm_GBs"] = hbm / sec * 1e-9
                e["hbm_frac_of_peak"] = e["hbm_GBs"] / HBM_PEAK_GBS
        out["kernels"][k] = e
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
