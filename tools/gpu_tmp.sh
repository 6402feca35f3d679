#!/bin/bash
# scratch script for one-off GPU experiments (rewritten per run): here, the round-end rehearsal
set -o pipefail
bash tools/gpu_final.sh tmp
