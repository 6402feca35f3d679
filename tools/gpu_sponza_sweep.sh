#!/bin/bash
# sponza stand-in: trace time against the number of tree nodes staged in LDS
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for n in 0 64 256 512 1024; do
  timeout -k 10 200 python bench.py --workload sponza --steps 5 --warmup 1 --no-cpu-baseline --no-kd --lds-nodes $n > $O/sp_$n.json 2> $O/sp_$n.err || { tail $O/sp_$n.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$O/sp_$n.json').read().strip().splitlines()[-1])
print('lds_nodes $n  Mrays/s %.0f  ms %.2f  trace %.2f shade %.2f' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['trace'], d['stage_ms_per_step']['shade']))"
done
