bash tools/gpu_pmc_all.sh f3_cb cornellbox.obj bvh && bash tools/gpu_pmc_all.sh f3_spb sponza bvh && bash tools/gpu_pmc_all.sh f3_spk sponza kd
