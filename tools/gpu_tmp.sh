bash tools/gpu_sweep.sh "--no-sponza --no-pmc --no-kd" --blocks-per-cu 3 4 5 6 8 && bash tools/gpu_sweep.sh "--no-sponza --no-pmc --no-kd" --tail-level 3 4 5 6 7
