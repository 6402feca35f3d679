#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/gpu_tests.txt
B="python3 bench.py --no-pmc --no-sponza --no-cpu-baseline --no-kd"
run() { name=$1; shift; timeout -k 10 300 $B "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])
print('%-26s %8.0f Mrays/s %7.3f ms  lanes %s tail rays %.3f' % ('$name', d['value'], d['ms_per_step'], d['config']['frames_in_flight'], d['roofline'].get('rays_in_fused_tail_fraction', -1)))"; }
run sp --workload sponza || exit 1
run spk --workload sponza --accel kd || exit 1
run tp512 --workload teapot --width 512 --height 384 || exit 1
run cb1024 || exit 1
