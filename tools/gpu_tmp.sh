#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
B="python3 bench.py --no-pmc --no-sponza --no-cpu-baseline --no-kd --steps 30 --warmup 5"
run() { name=$1; shift; timeout -k 10 300 $B "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])
print('%-26s %8.0f Mrays/s %7.3f ms  swaps %s' % ('$name', d['value'], d['ms_per_step'], d['config'].get('lane_streams_replaced')))"; }
for rep in 1 2; do
run s2_bvh || exit 1
run s3_bvh --streams 3 || exit 1
run s4_bvh --streams 4 || exit 1
run s2_kd --accel kd || exit 1
run s3_kd --accel kd --streams 3 || exit 1
run s2_coll --force-collective || exit 1
run s3_coll --force-collective --streams 3 || exit 1
done
run s2_teapot --workload teapot || exit 1
run s3_teapot --workload teapot --streams 3 || exit 1
run s2_alpha --workload cornellbox-alpha || exit 1
run s3_alpha --workload cornellbox-alpha --streams 3 || exit 1
run s2_glass --workload cornellbox-glass || exit 1
run s3_glass --workload cornellbox-glass --streams 3 || exit 1
run s2_sponza --workload sponza --steps 6 --warmup 2 || exit 1
run s3_sponza --workload sponza --steps 6 --warmup 2 --streams 3 || exit 1
echo rc=$?
