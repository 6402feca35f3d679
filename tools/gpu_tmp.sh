#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --force-collective --steps 10 --no-pmc --no-sponza --no-kd --no-cpu-baseline > gpurun_out/coll1.json 2> gpurun_out/coll1.err || { tail gpurun_out/coll1.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/coll1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_coll -- python3 $GRAFT_REPO_ROOT/bench.py --force-collective --steps 5 --warmup 1 --no-pmc --no-sponza --no-kd --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_coll.log 2>&1 || { tail $GRAFT_REPO_ROOT/gpurun_out/prof_coll.log; exit 2; }
find $GRAFT_REPO_ROOT/gpurun_out/prof_coll -name "*kernel_stats.csv" | head -1 | xargs cut -c1-90 | head -30
