#!/bin/bash
# SimpleMAE kernel tables after the key-padding free tiles
set -e -o pipefail
mkdir -p gpurun_out
B=32 bash tools/profile_cfg5.sh r04_k_cfg5_b32 > gpurun_out/r04_k_cfg5_b32.out 2>&1; tail -1 gpurun_out/r04_k_cfg5_b32.out
B=256 bash tools/profile_cfg5.sh r04_k_cfg5_b256 > gpurun_out/r04_k_cfg5_b256.out 2>&1; tail -1 gpurun_out/r04_k_cfg5_b256.out
