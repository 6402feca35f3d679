#!/bin/bash
# Sweep of the schedule's knobs on the workloads whose trees live in memory.  usage: gpu_sweep2.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
: > $O/sweep2_$T.txt
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-kd --no-sponza --no-pmc --steps 60 "$@" > $O/sw.json 2> $O/sw.err || { tail $O/sw.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('$O/sw.json').read().strip().splitlines()[-1])
print('%-70s %7.0f Mrays/s %7.3f ms  lanes %s' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step'], d['config'].get('frames_in_flight')))" "$@" | tee -a $O/sweep2_$T.txt
}
for w in "teapot --accel kd" "teapot --accel bvh" "bunny --accel bvh" "testobj --accel kd"; do
  run --workload $w
  run --workload $w --streams 4
  run --workload $w --blocks-per-cu 4
  run --workload $w --blocks-per-cu 6
  run --workload $w --lds-nodes 64
  run --workload $w --lds-nodes 256
  run --workload $w --tail-level 1
  run --workload $w --tail-level 3
  run --workload $w --traversal 0x0401
  run --workload $w --traversal 0x1001
  run --workload $w --traversal 0x21
  run --workload $w
done
