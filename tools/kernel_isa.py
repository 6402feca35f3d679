#!/usr/bin/env python3
"""Instruction audit of a trace-kernel instantiation: compiles trace_kernels.hip to gfx950 assembly (device side only, the
flags of the Makefile), finds the innermost loops of one kernel (a backward branch to a label with no other loop inside)
and prints, per loop, how many instructions of each kind its body holds: vector ALU, scalar ALU, LDS, vector memory,
selects (v_cndmask), three-operand min/max (v_min3 / v_max3 / v_med3), fused multiply-adds, reciprocals, waits.

usage: kernel_isa.py [--kernel SUBSTRING] [--list] [--dump LOOP_INDEX] [--extra "-DFOO"]
The default kernel is the one the headline workload runs: trace_kernel<KD=false, COUNT=false, SLAB_RECIP, DYNAMIC=true,
ALL_LDS=true, POOL=false> (mangled ...ILb0ELb0ELi1ELb1ELb1ELb0E...)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "path_tracer_amd", "csrc")


def assembly(source, extra):
    out = os.path.join(tempfile.mkdtemp(prefix="pt_isa_"), "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
           "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, source)] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def kernels(lines):
    """name -> (first line, last line) of every function in the listing."""
    out, cur, start = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, start = m.group(1), i
        elif l.startswith(".Lfunc_end") and cur:
            out[cur] = (start, i)
            cur = None
    return out


def classify(op):
    kinds = []
    if op.startswith("v_cndmask"):
        kinds.append("select")
    if re.match(r"v_(min3|max3|med3)", op):
        kinds.append("minmax3")
    if re.match(r"v_(min|max)_", op):
        kinds.append("minmax")
    if re.match(r"v_(fma|fmac|mad|pk_fma)", op):
        kinds.append("fma")
    if op.startswith("v_rcp"):
        kinds.append("rcp")
    if op.startswith("v_cmp"):
        kinds.append("vcmp")
    if op.startswith("v_div_"):
        kinds.append("div_seq")
    if op.startswith("s_waitcnt"):
        kinds.append("wait")
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        kinds.append("branch")
    if op.startswith("ds_"):
        kinds.append("lds")
    elif re.match(r"(global|buffer|flat|scratch)_", op):
        kinds.append("vmem")
    elif op.startswith("v_"):
        kinds.append("valu")
    elif op.startswith("s_"):
        kinds.append("salu")
    return kinds


def loops(body):
    """Innermost loops: (label line, branch line) pairs with no other backward branch target strictly inside."""
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            label_at[m.group(1)] = i
    back = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\w+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            back.append((label_at[m.group(1)], i))
    inner = [b for b in back if not any(o != b and b[0] <= o[0] and o[1] <= b[1] for o in back)]
    return sorted(set(inner)), sorted(set(back))


def count(body, lo, hi):
    tot = {}
    for l in body[lo:hi + 1]:
        m = re.match(r"\s+([a-z]\w+)", l)
        if not m or l.lstrip().startswith((";", ".")):
            continue
        for k in classify(m.group(1)):
            tot[k] = tot.get(k, 0) + 1
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--source", default="trace_kernels.hip")
    ap.add_argument("--kernel", default="trace_kernelILb0ELb0ELi1ELb1ELb1ELb0E")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--dump", type=int, default=None, help="print the instructions of loop number N")
    ap.add_argument("--all-loops", action="store_true", help="every loop, not only the innermost ones")
    ap.add_argument("--extra", default="")
    args = ap.parse_args()
    lines = assembly(args.source, args.extra.split())
    ks = kernels(lines)
    if args.list:
        for k in ks:
            print(k)
        return
    names = [k for k in ks if args.kernel in k]
    if len(names) != 1:
        sys.exit("kernel substring matches %d functions" % len(names))
    lo, hi = ks[names[0]]
    body = lines[lo:hi]
    meta = {}
    for l in lines[hi:hi + 400]:
        m = re.match(r"\s*;\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)\s*:?\s*(\d+)", l)
        if m:
            meta.setdefault(m.group(1), int(m.group(2)))
    print("kernel %s" % names[0])
    print("  " + "  ".join("%s %d" % kv for kv in meta.items()))
    whole = count(body, 0, len(body) - 1)
    print("  whole kernel: " + "  ".join("%s %d" % (k, whole[k]) for k in sorted(whole)))
    inner, every = loops(body)
    for n, (a, b) in enumerate(every if args.all_loops else inner):
        c = count(body, a, b)
        label = body[a].split(":")[0]
        print("  loop %d  %s  lines %d-%d (%d instructions): %s" % (
            n, label, a, b, sum(c.get(k, 0) for k in ("valu", "salu", "lds", "vmem")), "  ".join("%s %d" % (k, c[k]) for k in sorted(c))))
        if args.dump == n:
            for l in body[a:b + 1]:
                print("      " + l)


if __name__ == "__main__":
    main()
