#!/bin/bash
# One GPU-box pass: tests, bench, kernel stats, HBM-traffic counter passes.  usage: gpu_round.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/test_$T.log 2>&1 || { tail -30 $O/test_$T.log; exit 1; }
tail -2 $O/test_$T.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kd > $O/prof_$T.log 2>&1 || { tail $O/prof_$T.log; exit 2; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$T -- python3 $R/tools/prof_frame.py cornellbox.obj 1024 3 bvh > $O/pmcf_$T.log 2>&1 || { tail $O/pmcf_$T.log; exit 3; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$T -- python3 $R/tools/prof_frame.py cornellbox.obj 1024 3 bvh > $O/pmcw_$T.log 2>&1 || { tail $O/pmcw_$T.log; exit 4; }
bash $R/tools/gpu_pmc_sq.sh $T > /dev/null || exit 7
cp $O/pmc_valu_$T.json $R/profiles/r01_pmc_valu.json
cd $R
python3 tools/pmc_traffic.py $O/pmcf_$T $O/pmcw_$T $O/pmc_traffic_$T.json "cornellbox 1024x1024, 10 spp/frame, BVH, 3 frames" || exit 5
cp $O/pmc_traffic_$T.json profiles/r01_pmc_traffic.json
timeout -k 10 400 python bench.py --steps 30 --warmup 3 > $O/bench_$T.json 2> $O/bench_$T.err || { tail $O/bench_$T.err; exit 6; }
cat $O/bench_$T.json
