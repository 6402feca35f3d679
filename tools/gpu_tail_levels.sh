cd $GRAFT_REPO_ROOT
for tl in 255 0 2 3 4 6 8; do
  for a in bvh kd; do
  timeout -k 10 200 python bench.py --workload sponza --accel $a --tail-level $tl --no-cpu-baseline --no-kd --no-sponza --no-pmc --steps 8 --warmup 2 > gpurun_out/tl.json 2> gpurun_out/tl.err || { tail -3 gpurun_out/tl.err; exit 1; }
  python - $tl $a <<'PY'
import json,sys
d=json.load(open("gpurun_out/tl.json")); s=d["stage_ms_per_step"]
print("tail-level %s %s: %.0f Mrays/s %.2f ms/frame | one at a time: trace %.2f shade %.2f tail %.2f raygen %.2f accum %.2f | tail rays frac %.3f used %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], s["trace"], s["shade"], s["fused_tail"], s["raygen"], s["accumulate"], d["roofline"]["rays_in_fused_tail_fraction"], d["config"]["fused_tail_from_level"]))
PY
  done
done
