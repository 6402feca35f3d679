#!/usr/bin/env python3
"""Registers, LDS and private-segment bytes of every kernel in libpt_amd.so (from the code objects' metadata notes).
usage: kernel_regs.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("PT_AMD_LIB") or os.path.join(ROOT, "path_tracer_amd", "libpt_amd.so")
tmp = tempfile.mkdtemp()
subprocess.run(["cp", lib, tmp + "/l.so"], check=True)
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", "l.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
for f in sorted(os.listdir(tmp)):
    if "gfx950" not in f:
        continue
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(pt::SceneGpu.*", "", dem).replace("void pt::(anonymous namespace)::", "")
        if sys.argv[1:] and not any(a in dem for a in sys.argv[1:]):
            continue
        g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
        print("%-60s vgpr %3s sgpr %3s lds %6s scratch %5s" % (dem, g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
