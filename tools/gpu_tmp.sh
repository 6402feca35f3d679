#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 env PT_AMD_LIB=$PWD/ab_libs/libpt_lk.so python3 -u -m pytest tests/test_gpu_modes_and_cli.py -x -q --timeout 600 --timeout-method thread -k "traversal" > gpurun_out/tl_parity.txt 2>&1 || { tail -20 gpurun_out/tl_parity.txt; exit 1; }
tail -1 gpurun_out/tl_parity.txt
B="python3 bench.py --no-pmc --no-sponza --no-cpu-baseline --no-kd --steps 30 --warmup 5"
run() { name=$1; lib=$2; shift; shift; PT_AMD_LIB=$PWD/ab_libs/libpt_$lib.so timeout -k 10 300 $B "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])
print('%-26s %8.0f Mrays/s %7.3f ms  alone %.3f ms lanes %s' % ('$name', d['value'], d['ms_per_step'], d['roofline']['alone']['avg_launch_ms'], d['config']['frames_in_flight']))"; }
for rep in 1 2 3; do for r in base lk; do
run ${r}_bvh $r || exit 1
run ${r}_kd $r --accel kd || exit 1
run ${r}_teapot $r --workload teapot || exit 1
run ${r}_sponza $r --workload sponza --steps 6 --warmup 2 || exit 1
done; done
echo rc=$?
