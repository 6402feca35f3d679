#!/bin/bash
# usage: gpu_sweep.sh "<bench flags>" <flag> v1 v2 ...   -- one bench line per value of <flag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
base=$1; flag=$2; shift 2
for v in "$@"; do
  timeout -k 10 200 python bench.py $base $flag $v --steps 20 --warmup 3 --no-cpu-baseline > $O/sw.json 2> $O/sw.err || { tail $O/sw.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$O/sw.json').read().strip().splitlines()[-1])
print('$flag $v  Mrays/s %.0f  ms %.3f  %s  kd %s' % (d['value'], d['ms_per_step'], {k: round(x,3) for k,x in d['stage_ms_per_step'].items() if isinstance(x, float)}, round(d['kd_tree']['Mrays/s']) if 'kd_tree' in d else None))"
done
