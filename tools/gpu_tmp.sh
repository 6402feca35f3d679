#!/bin/bash
set -o pipefail
bash tools/gpu_final.sh f4
