#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/gpu_tmp.py || exit 1
for a in kd bvh; do
timeout -k 10 120 ./path_tracer_amd/CS488 tests/golden/scenes/cornellbox.obj --accel $a --frames 1000 --stats > gpurun_out/cli.txt 2> gpurun_out/cli.err; echo "$a: $(tail -1 gpurun_out/cli.err)"
done
timeout -k 10 120 ./path_tracer_amd/CS488 tests/golden/scenes/cornellbox.obj --accel bvh --gpus 1 --frames 300 --stats > gpurun_out/cli.txt 2> gpurun_out/cli.err; echo "group of 1: $(tail -1 gpurun_out/cli.err)"
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -30 gpurun_out/gpu_tests.txt; exit 1; }
tail -2 gpurun_out/gpu_tests.txt
timeout -k 10 300 python3 bench.py --no-pmc --no-cpu-baseline > gpurun_out/bench_def.json 2> gpurun_out/bench_def.err || { tail gpurun_out/bench_def.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_def.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['kd_tree'], {k:v['Mrays/s'] for k,v in d['sponza'].items() if isinstance(v,dict)})"
