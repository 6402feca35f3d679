#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="python3 bench.py --no-pmc --no-sponza --no-cpu-baseline --no-kd"
run() { name=$1; shift; timeout -k 10 300 $B "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])
print('%-26s %8.0f Mrays/s %7.3f ms  lanes %s' % ('$name', d['value'], d['ms_per_step'], d['config']['frames_in_flight']))"; }
run def || exit 1
run half --max-batch 5300000 || exit 1
run half_s3 --max-batch 5300000 --streams 3 || exit 1
run half_s2_t6 --max-batch 5300000 --streams 2 --tail-level 6 || exit 1
run half_s3_t4 --max-batch 5300000 --streams 3 --tail-level 4 || exit 1
run third --max-batch 3200000 || exit 1
run fifth --max-batch 2100000 || exit 1
run kd_def --accel kd || exit 1
run kd_half --accel kd --max-batch 5300000 || exit 1
run kd_half_s3_t4 --accel kd --max-batch 5300000 --streams 3 --tail-level 4 || exit 1
run sp_def --workload sponza || exit 1
run sp_half --workload sponza --max-batch 8400000 || exit 1
run sp_quarter --workload sponza --max-batch 4200000 || exit 1
