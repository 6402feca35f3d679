#!/bin/bash
# The CLI at the reference's own defaults (512 x 384, 10 spp), both accelerators, with and without the per-frame wait
# and with the counting kernels.  usage: gpu_cli_speed.sh
set -o pipefail
cd $GRAFT_REPO_ROOT
for args in "" "--accel bvh" "--drain-frames" "--accel bvh --drain-frames" "--count-tests" "--gpus 1" "--accel bvh --gpus 1"; do
  printf "%-32s " "CS488 cornellbox.obj $args"
  timeout -k 10 120 path_tracer_amd/CS488 tests/golden/scenes/cornellbox.obj --frames 1000 --stats $args 2>&1 >/dev/null | grep "spp/s" || exit 2
done
