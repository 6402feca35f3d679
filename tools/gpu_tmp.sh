cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/test_full.log 2>&1 || { tail -30 gpurun_out/test_full.log; exit 1; }
tail -2 gpurun_out/test_full.log
bash tools/gpu_r2_bench.sh b2
