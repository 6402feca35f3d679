cd $GRAFT_REPO_ROOT
for lib in path_tracer_amd/variants/libpt_amd_base.so "" path_tracer_amd/variants/libpt_amd_base.so ""; do
  PT_AMD_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-kd --no-sponza --no-pmc --steps 100 --warmup 5 > gpurun_out/st.json 2> gpurun_out/st.err || { tail -3 gpurun_out/st.err; exit 1; }
  python - "$lib" <<'PY'
import json,sys
d=json.load(open("gpurun_out/st.json")); s=d["stage_ms_per_step"]
print("%-45s %.0f Mrays/s %.3f ms | trace %.3f shade %.3f tail %.3f raygen %.3f accum %.3f" % (sys.argv[1] or "this build", d["value"], d["ms_per_step"], s["trace"], s["shade"], s["fused_tail"], s["raygen"], s["accumulate"]))
PY
done
