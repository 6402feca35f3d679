cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/coherence_probe.py sponza 1024 bvh > gpurun_out/coh_sponza_bvh.txt 2>&1 || { tail gpurun_out/coh_sponza_bvh.txt; exit 1; }
timeout -k 10 200 python tools/coherence_probe.py sponza 1024 kd > gpurun_out/coh_sponza_kd.txt 2>&1 || { tail gpurun_out/coh_sponza_kd.txt; exit 1; }
timeout -k 10 200 python tools/coherence_probe.py teapot.obj 1024 kd > gpurun_out/coh_teapot_kd.txt 2>&1 || { tail gpurun_out/coh_teapot_kd.txt; exit 1; }
cat gpurun_out/coh_sponza_bvh.txt gpurun_out/coh_sponza_kd.txt gpurun_out/coh_teapot_kd.txt
