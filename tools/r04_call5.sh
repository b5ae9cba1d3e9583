#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
bash tools/profile_step.sh r04_e "round 4, nt GEMM stores + scratch-free forward" > gpurun_out/r04_e_profile.out 2>&1; tail -1 gpurun_out/r04_e_profile.out | cut -c1-200
bash tools/pmc_step.sh r04_e > gpurun_out/r04_e_pmc_step.out 2>&1; tail -3 gpurun_out/r04_e_pmc_step.out
B=32 bash tools/profile_cfg5.sh r04_e_cfg5_b32 > gpurun_out/r04_e_cfg5_b32.out 2>&1; tail -1 gpurun_out/r04_e_cfg5_b32.out
B=256 bash tools/profile_cfg5.sh r04_e_cfg5_b256 > gpurun_out/r04_e_cfg5_b256.out 2>&1; tail -1 gpurun_out/r04_e_cfg5_b256.out
timeout -k 10 300 python tools/other_configs_bench.py > gpurun_out/r04_e_other_configs.txt 2>&1; cat gpurun_out/r04_e_other_configs.txt
