bash tools/gpu_pmc_all.sh f1_cb cornellbox.obj bvh
