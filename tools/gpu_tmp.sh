cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_converged.py -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/test_conv.log 2>&1; rc=$?; tail -5 gpurun_out/test_conv.log; exit $rc
