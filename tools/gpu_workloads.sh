#!/bin/bash
# One bench line per named workload (both accelerators) -> gpurun_out/workloads_<tag>.jsonl
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd $R
: > $O/workloads_$T.jsonl
for w in cornellbox cornellbox-glass cornellbox-metal cornellbox-alpha cornellball-glass teapot teapot-glass testobj bunny sponza; do
  for a in bvh kd; do
    STEPS="--steps 60 --warmup 5"; [ $w = sponza ] && STEPS=""
    timeout -k 10 300 python bench.py --workload $w --accel $a $STEPS --no-cpu-baseline --no-kd --no-pmc > $O/wl.json 2> $O/wl.err || { tail $O/wl.err; exit 1; }
    tail -1 $O/wl.json >> $O/workloads_$T.jsonl
    python3 -c "
import json
d=json.loads(open('$O/wl.json').read().strip().splitlines()[-1]); r=d['roofline']
print('%-18s %-3s %7.0f Mrays/s %8.1f spp/s %8.3f ms  rays/sample %.2f box %.1f tri %.1f' % ('$w', '$a', d['value'], d['spp_per_s'], d['ms_per_step'], d['rays_per_sample'], r['box_tests_per_ray'], r['tri_tests_per_ray']))"
  done
done
