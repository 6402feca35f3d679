bash tools/gpu_r2_kd.sh k4 kd
