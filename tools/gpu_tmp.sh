#!/bin/bash
set -o pipefail
bash tools/gpu_pmc_all.sh cb cornellbox.obj bvh && bash tools/gpu_pmc_all.sh spb sponza bvh && bash tools/gpu_pmc_all.sh spk sponza kd && bash tools/gpu_final.sh f6
echo rc=$?
