cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -u -m pytest tests/test_gpu_config_scenes.py -k sponza -m gpu -x -v --timeout 600 --timeout-method thread > gpurun_out/test_cfg.log 2>&1; rc=$?; tail -30 gpurun_out/test_cfg.log; exit $rc
