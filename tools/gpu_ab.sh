#!/bin/bash
# A/B of trace-kernel settings within one GPU-box call: every workload x every setting, twice, alternating.
# usage: gpu_ab.sh <tag> "<workload> ..." "<bench flags A>" "<bench flags B>" [more settings ...]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
W=$1; shift
cd $R
: > $O/ab_$T.txt
for round in 1 2; do
  for w in $W; do
    for accel in ${ACCELS:-bvh}; do
      i=0
      for flags in "$@"; do
        i=$((i+1))
        # a setting of the form "LIB=<file> <flags>" runs the flags on another build of the library (path_tracer_amd/variants/)
        # "ENV=<NAME=value>[,<NAME=value>...] <flags>" sets environment variables for the run
        envs=""
        case "$flags" in ENV=*) envs="${flags%% *}"; envs="${envs#ENV=}"; envs="${envs//,/ }"; flags="${flags#* }";; esac
        lib=""; f="$flags"
        case "$flags" in LIB=*) lib="$R/path_tracer_amd/variants/${flags%% *}"; lib="${lib/LIB=/}"; f="${flags#* }"; [ "$f" = "$flags" ] && f="";; esac
        env $envs PT_AMD_LIB=$lib timeout -k 10 150 python bench.py --workload $w --accel $accel --no-cpu-baseline --no-kd --no-sponza --no-pmc ${STEPS:---steps 40 --warmup 5} $f > $O/ab_tmp.json 2> $O/ab_tmp.err || { tail -5 $O/ab_tmp.err; exit 1; }
        python - "$w" "$accel" "$i" "$flags" $O/ab_tmp.json >> $O/ab_$T.txt <<'PY'
import json, sys
w, accel, i, flags, path = sys.argv[1:6]
r = json.load(open(path))
print("%-18s %-4s setting %s [%s]  %9.1f Mrays/s  %8.3f ms/frame  trace alone %.3f ms/launch" % (
    w, accel, i, flags, r["value"], r["ms_per_step"], r["roofline"]["alone"]["avg_launch_ms"]))
PY
      done
    done
  done
done
cat $O/ab_$T.txt
