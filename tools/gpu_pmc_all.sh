#!/bin/bash
# Every counter pass of one workload, each in a run of its own: kernel trace, two SQ passes, HBM fetch, HBM write,
# L2 (TCC), L1 (TCP).  usage: gpu_pmc_all.sh <tag> <scene: cornellbox.obj | teapot.obj | sponza> <bvh|kd> [traversal mode]
# (PT_PROF_TAIL_LEVEL in the environment is passed on to the workload)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; S=${2:-cornellbox.obj}; A=${3:-bvh}
cd /tmp && export TMPDIR=/tmp
W="python3 $R/tools/prof_frame.py $S 1024 3 $A $4"
pass() {  # name, rocprofv3 arguments...
    local name=$1; shift
    timeout -s KILL 150 rocprofv3 "$@" --output-format csv -d $O/${name}_$T -- $W > $O/${name}_$T.log 2>&1 || { tail -5 $O/${name}_$T.log; exit 1; }
}
pass sqt --kernel-trace --stats
pass sq1 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU
pass sq2 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FLOPS_FP32 SQ_ACTIVE_INST_SCA
pass pmcf --pmc FETCH_SIZE
pass pmcw --pmc WRITE_SIZE
pass tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum
cd $R
python3 tools/pmc_valu.py $O/sq1_$T $O/sq2_$T --trace $O/sqt_$T --out $O/pmc_valu_$T.json --note "$S, $A, 3 frames, one in flight" > /dev/null || exit 4
python3 tools/pmc_cache.py $O/pmcf_$T $O/pmcw_$T $O/tcc_$T $O/tcp_$T --trace $O/sqt_$T --out $O/pmc_cache_$T.json --note "$S, $A, 3 frames, one in flight" > /dev/null || exit 5
find $O/sqt_$T -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$T.csv
python3 - <<PY
import json
v=json.load(open("$O/pmc_valu_$T.json"))["kernels"]; c=json.load(open("$O/pmc_cache_$T.json"))["kernels"]
for k in v:
    e=v[k]; f=c.get(k,{})
    print("%-18s %7.1f us  valu %.2f salu %.2f lanes %.2f | L1 hit %s L2 hit %s L2 GB/s<= %s HBM GB/s %s" % (k, e["avg_launch_s"]*1e6, e.get("valu_issue_busy",0), e.get("salu_issue_busy",0), e.get("valu_lane_occupancy",0),
        "%.2f"%f["l1_read_hit_rate"] if "l1_read_hit_rate" in f else "-", "%.2f"%f["l2_hit_rate"] if "l2_hit_rate" in f else "-", "%.0f"%f["l2_GBs_upper_bound"] if "l2_GBs_upper_bound" in f else "-", "%.0f"%f["hbm_GBs"] if "hbm_GBs" in f else "-"))
PY
