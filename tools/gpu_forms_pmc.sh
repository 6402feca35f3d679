#!/bin/bash
# Counter passes of the three forms of the walk in memory on the stand-in (one trace launch per level, workgroups on
# their own regions): 64-byte nodes, 192-byte nodes (traversal bit 15), two rays per lane (bit 1).
# usage: gpu_forms_pmc.sh <bvh|kd>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; A=${1:-bvh}
cd $R
export PT_PROF_TAIL_LEVEL=0
bash tools/gpu_pmc_all.sh form_plain_$A sponza $A 0x09 > $O/form_plain_$A.txt 2>&1 || { tail $O/form_plain_$A.txt; exit 1; }
if [ "$A" = bvh ]; then bash tools/gpu_pmc_all.sh form_wide_$A sponza $A 0x8009 > $O/form_wide_$A.txt 2>&1 || { tail $O/form_wide_$A.txt; exit 2; }; fi
bash tools/gpu_pmc_all.sh form_two_$A sponza $A 0x0b > $O/form_two_$A.txt 2>&1 || { tail $O/form_two_$A.txt; exit 3; }
python3 - "$A" <<'PY'
import json, sys, os
A = sys.argv[1]
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
rows = [("64-byte nodes, one ray a lane", "plain"), ("192-byte nodes (bit 15)", "wide"), ("two rays a lane (bit 1)", "two")]
print("stand-in, %s, 3840x2160, 4 spp, one batch at a time, one trace launch per level; trace kernel per launch (average of %s)" % (A, "all levels"))
print("%-32s %8s %9s %6s %6s %9s %9s %7s %9s %7s %9s" % ("form", "ms", "VALU M", "busy", "lanes", "VMEM rd M", "L1 acc M", "L1 hit", "L2 req M", "L2 hit", "fabric GB"))
for name, key in rows:
    fv, fc = "%s/pmc_valu_form_%s_%s.json" % (O, key, A), "%s/pmc_cache_form_%s_%s.json" % (O, key, A)
    if not os.path.exists(fv):
        continue
    v = json.load(open(fv))["kernels"]["trace_kernel"]; c = json.load(open(fc))["kernels"]["trace_kernel"]
    pv, pc = v["per_launch"], c["per_launch"]
    print("%-32s %8.3f %9.1f %6.2f %6.2f %9.2f %9.1f %7.3f %9.1f %7.3f %9.2f" % (
        name, v["avg_launch_s"] * 1e3, pv["SQ_INSTS_VALU"] / 1e6, v["valu_issue_busy"], v["valu_lane_occupancy"], pv["SQ_INSTS_VMEM_RD"] / 1e6,
        pc["TCP_TOTAL_CACHE_ACCESSES_sum"] / 1e6, c["l1_read_hit_rate"], pc["TCC_REQ_sum"] / 1e6, c["l2_hit_rate"], c["hbm_bytes_per_launch"] / 1e9))
PY
