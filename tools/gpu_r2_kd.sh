#!/bin/bash
# Parity of both walks, then kd-tree and BVH throughput on three workloads.  usage: gpu_r2_kd.sh <tag> [accels]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; ACC=${2:-"kd bvh"}
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_trace_parity.py tests/test_gpu_modes_and_cli.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/test_$T.log 2>&1 || { tail -40 $O/test_$T.log; exit 1; }
tail -2 $O/test_$T.log
for w in cornellbox teapot sponza; do for a in $ACC; do
timeout -k 10 200 python bench.py --workload $w --accel $a --steps 12 --no-cpu-baseline --no-sponza --no-pmc --no-kd > $O/bench_${w}_${a}_$T.json 2> $O/bench_${w}_${a}_$T.err || { tail $O/bench_${w}_${a}_$T.err; exit 2; }
python3 - <<PY
import json
d=json.loads(open("$O/bench_${w}_${a}_$T.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$w $a Mrays/s %.0f ms %.3f alone_ms %.3f fp32 %.3f box/ray %.1f tri/ray %.1f stages %s" % (d["value"], d["ms_per_step"], r["alone"]["avg_launch_ms"], r["fractions"]["valu"], r["box_tests_per_ray"], r["tri_tests_per_ray"], {k:round(v,2) for k,v in d["stage_ms_per_step"].items() if k!="note"}))
PY
done; done
