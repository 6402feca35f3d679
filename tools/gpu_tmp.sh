#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/gpu_tests.txt
for a in kd bvh; do for extra in "" "--drain-frames"; do
timeout -k 10 120 ./path_tracer_amd/CS488 tests/golden/scenes/cornellbox.obj --accel $a --frames 400 --stats $extra > gpurun_out/cli.txt 2> gpurun_out/cli.err; echo "$a $extra: $(tail -1 gpurun_out/cli.err)"
done; done
B="python3 bench.py --no-pmc --no-sponza --no-cpu-baseline --no-kd"
run() { name=$1; shift; timeout -k 10 300 $B "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || return 1; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$name.json').read().strip().splitlines()[-1])
print('%-26s %8.0f Mrays/s %7.3f ms  spp/s %.0f lanes %s' % ('$name', d['value'], d['ms_per_step'], d['spp_per_s'], d['config']['frames_in_flight']))"; }
run r1024 || exit 1
run r512x384 --width 512 --height 384 || exit 1
run r512x384_kd --width 512 --height 384 --accel kd || exit 1
