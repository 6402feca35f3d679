#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/gpu_tests.txt
