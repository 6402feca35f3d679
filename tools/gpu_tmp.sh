cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_raygen.py tests/test_gpu_trace_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/test_rg.log 2>&1; rc=$?; tail -30 gpurun_out/test_rg.log; exit $rc
