#!/bin/bash
set -o pipefail
bash tools/gpu_pmc_all.sh cb cornellbox.obj bvh && bash tools/gpu_pmc_all.sh spb sponza bvh && bash tools/gpu_pmc_all.sh spk sponza kd
echo rc=$?
