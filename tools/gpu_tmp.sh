#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PT_AMD_LIB=$PWD/ab_libs/libpt_timeline.so
{
echo "== workgroups on their own regions (PT_OPT_TRAVERSAL bit 3), BVH"; timeout -k 10 200 python3 tools/wg_timeline.py cornellbox.obj bvh 5800000 OPT_TRAVERSAL=9 | grep -v "^xcd" || exit 1
echo; echo "== shared pools (pt_trace_batch's default), BVH"; timeout -k 10 200 python3 tools/wg_timeline.py cornellbox.obj bvh 5800000 | grep -v "^xcd" || exit 1
echo; echo "== own regions, kd-tree"; timeout -k 10 200 python3 tools/wg_timeline.py cornellbox.obj kd 5800000 OPT_TRAVERSAL=9 | grep -v "^xcd" || exit 1
echo; echo "== shared pools, kd-tree"; timeout -k 10 200 python3 tools/wg_timeline.py cornellbox.obj kd 5800000 | grep -v "^xcd" || exit 1
} > gpurun_out/wg_timeline.txt 2>&1
echo rc=$?; grep -E "==|launch by|wave-time|at the peak" gpurun_out/wg_timeline.txt
