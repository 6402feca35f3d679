#!/bin/bash
# Kernel trace of the CLI at the reference's own defaults (512 x 384, 10 spp, kd-tree).  usage: gpu_cli_trace.sh <tag> [extra CLI args]
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli_$T -- $R/path_tracer_amd/CS488 $R/tests/golden/scenes/cornellbox.obj --frames 300 --stats "$@" > $O/cli_$T.log 2>&1 || { tail $O/cli_$T.log; exit 1; }
grep "spp/s" $O/cli_$T.log
find $O/cli_$T -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cli_stats_$T.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/cli_stats_$T.csv")))
for r in rows[:9]:
    n=r["Name"]; n=n[n.find("::",n.find("namespace"))+2:][:60] if "namespace" in n else n[:60]
    print("%-62s calls %5s avg %8.1f us  %5s %%" % (n, r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
