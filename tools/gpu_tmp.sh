cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_image_parity.py tests/test_gpu_modes_and_cli.py tests/test_gpu_group.py tests/test_gpu_config_scenes.py -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/test_fx.log 2>&1 || { tail -30 gpurun_out/test_fx.log; exit 1; }
tail -2 gpurun_out/test_fx.log
bash tools/gpu_sweep.sh "--no-sponza --no-pmc --no-kd --workload cornellbox-alpha" --streams 2
bash tools/gpu_sweep.sh "--no-sponza --no-pmc --no-kd --workload cornellbox-alpha --accel kd" --streams 2
bash tools/gpu_sweep.sh "--no-sponza --no-pmc --no-kd --workload sponza" --streams 2
