cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload sponza --strong --spp 64 --force-collective --steps 2 --warmup 1 --no-pmc > gpurun_out/bench_sp_strong.json 2> gpurun_out/bench_sp_strong.err || { tail gpurun_out/bench_sp_strong.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_sp_strong.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['spp_per_s'], d['scaling'], d['config'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['fractions'])"
timeout -k 10 300 ./path_tracer_amd/CS488 gpurun_out/sponza/sponza_standin.obj tests/golden/scenes/probe_64.hdr --gpus 1 --accel bvh --width 1920 --height 1080 --spp 8 --frames 3 --stats --eye -2.0 0.05 0.15 --lookat 1.0 -0.1 -0.05 --out gpurun_out/sponza_cli.png | tail -8
