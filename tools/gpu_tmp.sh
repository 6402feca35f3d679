#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
s=$(date +%s)
timeout -k 10 600 python3 bench.py > gpurun_out/bench_def.json 2> gpurun_out/bench_def.err || { tail gpurun_out/bench_def.err; exit 1; }
e=$(date +%s); echo "default bench wall $((e-s)) s"
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_def.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], r['frac'], r['avg_launch_ms'], r['alone']['frac'], d['kd_tree'], {k:v['Mrays/s'] for k,v in d['sponza'].items() if isinstance(v,dict)}, d['cpu_baseline']['value'])"
timeout -k 10 300 python3 bench.py --workload sponza --no-pmc --no-cpu-baseline > gpurun_out/bench_sp.json 2> gpurun_out/bench_sp.err || { tail gpurun_out/bench_sp.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_sp.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['steps'], r['bound'], r['frac'], r['fractions'])"
