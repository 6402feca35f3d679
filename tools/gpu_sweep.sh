#!/bin/bash
# Sweep of the schedule's knobs around their defaults on the default workload (both accelerators), one bench line each.
# usage: gpu_sweep.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
: > $O/sweep_$T.txt
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-kd --no-sponza --no-pmc --steps 100 "$@" > $O/sw.json 2> $O/sw.err || { tail $O/sw.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('$O/sw.json').read().strip().splitlines()[-1])
print('%-60s %7.0f Mrays/s %7.3f ms  lanes %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['config'].get('frames_in_flight')))" "$@" | tee -a $O/sweep_$T.txt
}
for a in bvh kd; do
  run --accel $a
  run --accel $a --streams 3
  run --accel $a --streams 2 --blocks-per-cu 4
  run --accel $a --streams 2 --blocks-per-cu 6
  run --accel $a --tail-level 5
  run --accel $a --tail-level 4
  run --accel $a --max-batch 5242880
  run --accel $a
done
