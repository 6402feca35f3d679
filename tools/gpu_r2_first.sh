#!/bin/bash
# Round-2 first pass: the new allocation-free entry, the whole GPU suite, the default bench.  usage: gpu_r2_first.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -m gpu -x -v --timeout 120 --timeout-method thread > $O/noalloc_$T.log 2>&1 || { tail -30 $O/noalloc_$T.log; exit 1; }
tail -3 $O/noalloc_$T.log
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/test_$T.log 2>&1 || { tail -30 $O/test_$T.log; exit 2; }
tail -2 $O/test_$T.log
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_$T.json 2> $O/bench_$T.err || { tail $O/bench_$T.err; exit 6; }
python3 - <<PY
import json
d=json.loads(open("$O/bench_$T.json").read().strip().splitlines()[-1])
print("Mrays/s %.0f  ms/frame %.3f  stages %s  kd %s" % (d["value"], d["ms_per_step"], d["stage_ms_per_step"], d.get("kd_tree")))
PY
