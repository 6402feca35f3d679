#!/bin/bash
# Round-end rehearsal: the whole GPU suite, smoke(), the default bench, the one-rank collective runs, and the kernel
# trace of the bench.  usage: gpu_final.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > $O/test_$T.log 2>&1 || { tail -30 $O/test_$T.log; exit 1; }
tail -2 $O/test_$T.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$T.log 2>&1 || { tail $O/smoke_$T.log; exit 2; }
tail -1 $O/smoke_$T.log
bash tools/gpu_r2_bench.sh $T
