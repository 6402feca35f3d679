#!/bin/bash
# Default bench (live counter passes, sponza sub-record, CPU baseline), the one-rank collective path, and the
# kernel trace of the bench itself.  usage: gpu_r2_bench.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
timeout -k 10 500 python bench.py > $O/bench_$T.json 2> $O/bench_$T.err || { tail $O/bench_$T.err; exit 1; }
timeout -k 10 200 python bench.py --force-collective --no-pmc --no-cpu-baseline --no-sponza > $O/bench_coll_$T.json 2> $O/bench_coll_$T.err || { tail $O/bench_coll_$T.err; exit 2; }
timeout -k 10 200 python bench.py --force-collective --strong --workload cornellbox-alpha --steps 60 --no-pmc > $O/bench_coll_alpha_$T.json 2> $O/bench_coll_alpha_$T.err || { tail $O/bench_coll_alpha_$T.err; exit 3; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kd --no-sponza --no-pmc > $O/prof_$T.log 2>&1 || { tail $O/prof_$T.log; exit 4; }
find $O/prof_$T -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench_$T.csv
cd $R
python3 - <<PY
import json
for f in ("bench_$T","bench_coll_$T","bench_coll_alpha_$T"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "Mrays/s %.0f ms %.3f bound %s frac %.3f fracs %s traffic %s" % (d["value"], d["ms_per_step"], r["bound"], r["frac"], r["fractions"], r["traffic"]))
    if "sponza" in d: print(" sponza", {k:(v["Mrays/s"], v["trace_kernel_alone"]["fp32_frac"]) for k,v in d["sponza"].items() if isinstance(v,dict)})
    if "cpu_baseline" in d: print(" cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["threads"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_thread_per_core"]["value"], d["cpu_baseline"]["host"])
PY
