#!/bin/bash
# Round-3 evidence pass on the GPU box: the whole GPU suite, smoke(), the default bench (as the driver runs it), the
# multi-rank launch paths (self-launched and under torch.distributed.run, gloo on the box's one GPU; one-rank RCCL),
# the kernel trace of the bench, and the CS488 host at the reference's settings.  usage: gpu_r3.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > $O/test_$T.log 2>&1 || { tail -30 $O/test_$T.log; exit 1; }
tail -2 $O/test_$T.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$T.log 2>&1 || { tail $O/smoke_$T.log; exit 2; }
tail -1 $O/smoke_$T.log
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_$T.json 2> $O/bench_$T.err || { tail $O/bench_$T.err; exit 3; }
timeout -k 10 500 python bench.py --steps 200 --warmup 10 --no-sponza --no-cpu-baseline > $O/bench_long_$T.json 2> $O/bench_$T.err || { tail $O/bench_$T.err; exit 3; }
# no launcher, WORLD_SIZE unset: the bench starts its two ranks itself (both on the box's one GPU, gloo)
env -u WORLD_SIZE -u RANK -u LOCAL_RANK timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 30 --warmup 5 --no-pmc > $O/bench_gloo2_self_$T.json 2> $O/bench_gloo2_self_$T.err || { tail $O/bench_gloo2_self_$T.err; exit 4; }
# the same under the launcher the task names
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 30 --warmup 5 --no-pmc > $O/bench_gloo2_torchrun_$T.json 2> $O/bench_gloo2_torchrun_$T.err || { tail $O/bench_gloo2_torchrun_$T.err; exit 5; }
# one rank through RCCL, a reduce every fourth frame
timeout -k 10 300 python bench.py --force-collective --reduce-every 4 --no-pmc --no-cpu-baseline --no-sponza > $O/bench_coll_$T.json 2> $O/bench_coll_$T.err || { tail $O/bench_coll_$T.err; exit 6; }
timeout -k 10 300 python bench.py --force-collective --reduce-every 1 --no-pmc --no-cpu-baseline --no-sponza > $O/bench_coll1_$T.json 2> $O/bench_coll_$T.err || { tail $O/bench_coll_$T.err; exit 6; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kd --no-sponza --no-pmc > $O/prof_$T.log 2>&1 || { tail $O/prof_$T.log; exit 7; }
find $O/prof_$T -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench_$T.csv
cd $R
bash tools/gpu_cli_speed.sh > $O/cli_speed_$T.txt 2>&1 || { tail $O/cli_speed_$T.txt; exit 8; }
python3 - <<PY
import json
for f in ("bench_driver_$T","bench_long_$T","bench_gloo2_self_$T","bench_gloo2_torchrun_$T","bench_coll_$T","bench_coll1_$T"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "n_gpus %d Mrays/s %.0f ms %.3f frac %.3f alone %.3f kernel_ms %.3f cfg %s" % (d["n_gpus"], d["value"], d["ms_per_step"], r["frac"], r["alone"]["frac"], r["kernel_ms_per_step"], {k:d["config"].get(k) for k in ("reduce_every","collective_ranks","collective_backend","self_launched")}), d.get("per_rank_Mrays/s"))
    if "sponza" in d: print(" sponza", {k:(v.get("Mrays/s"), v.get("gpu_over_cpu")) for k,v in d["sponza"].items() if isinstance(v,dict)})
    if "config1" in d: print(" config1", d["config1"]["gpu"], d["config1"].get("gpu_over_cpu"))
    if "cpu_baseline" in d: print(" cpu", d["cpu_baseline"]["value"], d.get("gpu_over_cpu"))
PY
tail -12 $O/cli_speed_$T.txt
