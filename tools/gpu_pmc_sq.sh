#!/bin/bash
# VALU / LDS counter passes over three cornellbox frames, plus a kernel trace of the same program.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1
cd /tmp && export TMPDIR=/tmp
W="python3 $R/tools/prof_frame.py ${2:-cornellbox.obj} 1024 3 bvh"
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $O/sqt_$T -- $W > $O/sqt_$T.log 2>&1 || { tail $O/sqt_$T.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/sq1_$T -- $W > $O/sq1_$T.log 2>&1 || { tail $O/sq1_$T.log; exit 2; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FLOPS_FP32 SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2_$T -- $W > $O/sq2_$T.log 2>&1 || { tail $O/sq2_$T.log; exit 3; }
cd $R
python3 tools/pmc_valu.py $O/sq1_$T $O/sq2_$T --trace $O/sqt_$T --out $O/pmc_valu_$T.json --note "${2:-cornellbox.obj}, BVH, 3 frames" > $O/pmc_valu_$T.txt || exit 4
cat $O/pmc_valu_$T.txt
