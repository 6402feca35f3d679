bash tools/gpu_r2_kd.sh q1 kd
