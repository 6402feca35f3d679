#!/bin/bash
# Builds another copy of libpt_amd.so with extra kernel flags into path_tracer_amd/variants/libpt_amd_<name>.so, for
# A/B runs within one GPU-box call (tools/gpu_ab.sh "LIB=libpt_amd_<name>.so <bench flags>"; path_tracer_amd reads
# PT_AMD_LIB).  The measurement flags are described where they are used in csrc/trace_device.h:
#   -DPT_PROBE_EXTRA_FETCH=n [-DPT_PROBE_FETCH_DWORD] [-DPT_PROBE_FETCH_ONE_LANE]   n more loads per step of a node in memory
#   -DPT_PROBE_EXTRA_VALU=n / -DPT_PROBE_EXTRA_SALU=n / -DPT_PROBE_EXTRA_LDS=n       n more vector / scalar / LDS-read instructions per node step
#   -DPT_BVH_STACK_ROWS=n / -DPT_KD_STACK_ROWS=n                                      LDS rows of a lane's traversal stack (defaults 16 / 8; on the Cornell boxes 12 / 6 and 10 / 5 change nothing)
#   -DPT_TWO_STACK_ROWS_BVH=n / -DPT_TWO_STACK_ROWS_KD=n                             LDS stack rows per ray of the two-rays-per-lane form (8 / 6)
# usage: build_variant.sh <name> "<flags>"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; F=$2
W=$(mktemp -d /tmp/pt_variant_XXXX)
mkdir -p $W/a/b/csrc $W/a/include $R/path_tracer_amd/variants
cp $R/path_tracer_amd/csrc/*.cpp $R/path_tracer_amd/csrc/*.h $R/path_tracer_amd/csrc/*.hip $R/path_tracer_amd/csrc/Makefile $W/a/b/csrc/
cp $R/include/*.h $W/a/include/
make -C $W/a/b/csrc EXTRA="$F" ../libpt_amd.so > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cp $W/a/b/libpt_amd.so $R/path_tracer_amd/variants/libpt_amd_$N.so
rm -rf $W
echo "path_tracer_amd/variants/libpt_amd_$N.so"
