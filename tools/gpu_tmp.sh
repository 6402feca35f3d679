cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_group.py -x -v --timeout 300 --timeout-method thread > gpurun_out/test_grp.log 2>&1; rc=$?; tail -40 gpurun_out/test_grp.log; exit $rc
