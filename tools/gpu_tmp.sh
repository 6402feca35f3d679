#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="python3 bench.py --no-pmc --no-sponza --no-cpu-baseline --no-kd --steps 30 --warmup 5"
run() { name=$1; lib=$2; shift; shift; PT_AMD_LIB=$PWD/ab_libs/libpt_$lib.so timeout -k 10 300 $B "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])
print('%-22s %8.0f Mrays/s %7.3f ms' % ('$name', d['value'], d['ms_per_step']))"; }
for r in noattr sgpr72; do
PT_AMD_LIB=$PWD/ab_libs/libpt_$r.so timeout -k 10 200 python3 tools/wg_timeline.py cornellbox.obj bvh 5800000 OPT_TRAVERSAL=9 > gpurun_out/tl_$r.txt 2>&1 || exit 1
grep -E "launch by|waves running" gpurun_out/tl_$r.txt
PT_AMD_LIB=$PWD/ab_libs/libpt_$r.so timeout -k 10 200 python3 tools/wg_timeline.py cornellbox.obj kd 5800000 OPT_TRAVERSAL=9 > gpurun_out/tl_kd_$r.txt 2>&1 || exit 1
grep -E "launch by|waves running" gpurun_out/tl_kd_$r.txt
done
for rep in 1 2; do for r in noattr sgpr72; do
run $r $r || exit 1
run ${r}_bpc6 $r --blocks-per-cu 6 || exit 1
run ${r}_bpc8 $r --blocks-per-cu 8 || exit 1
run ${r}_s1 $r --streams 1 || exit 1
run ${r}_kd $r --accel kd || exit 1
done; done
for r in noattr sgpr72; do
run ${r}_teapot $r --workload teapot || exit 1
run ${r}_teapot_kd $r --workload teapot --accel kd || exit 1
run ${r}_sponza $r --workload sponza --steps 6 --warmup 2 || exit 1
run ${r}_sponza_kd $r --workload sponza --accel kd --steps 4 --warmup 1 || exit 1
done
echo rc=$?
