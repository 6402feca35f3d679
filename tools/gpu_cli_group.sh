#!/bin/bash
# Group tests, then the CLI frame loop with and without a group and with and without the per-frame wait.
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_group.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_group.log 2>&1 || { tail -20 gpurun_out/t_group.log; exit 1; }
tail -2 gpurun_out/t_group.log
for args in "" "--gpus 1" "--gpus 1 --drain-frames" "--drain-frames"; do
  echo "CS488 cornellbox.obj $args"
  timeout -k 10 120 path_tracer_amd/CS488 tests/golden/scenes/cornellbox.obj --frames 400 --stats $args 2>&1 >/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -1 || exit 2
done
