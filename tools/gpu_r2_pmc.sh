#!/bin/bash
# Whole GPU suite (the allocation-free entry's test runs last), then every counter pass for three workloads.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/noalloc_$T.log 2>&1 || { tail -30 $O/noalloc_$T.log; exit 1; }
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/test_$T.log 2>&1 || { tail -30 $O/test_$T.log; exit 2; }
tail -2 $O/test_$T.log
bash tools/gpu_pmc_all.sh ${T}_cb cornellbox.obj bvh || exit 3
bash tools/gpu_pmc_all.sh ${T}_spb sponza bvh || exit 4
bash tools/gpu_pmc_all.sh ${T}_spk sponza kd || exit 5
