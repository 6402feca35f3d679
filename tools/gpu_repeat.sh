#!/bin/bash
# The default bench n times in fresh processes: spread of the result.  usage: gpu_repeat.sh <n> [bench flags]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
n=$1; shift
for i in $(seq 1 $n); do
  timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-kd "$@" > $O/rep.json 2> $O/rep.err || { tail $O/rep.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$O/rep.json').read().strip().splitlines()[-1])
print('run $i  Mrays/s %.0f  ms %.3f  replaced %s' % (d['value'], d['ms_per_step'], d['config'].get('lane_streams_replaced')))"
done
