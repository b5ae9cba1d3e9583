#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
B=32 bash tools/profile_cfg5.sh r04_g_cfg5_b32 > gpurun_out/r04_g_cfg5_b32.out 2>&1; tail -1 gpurun_out/r04_g_cfg5_b32.out
B=256 bash tools/profile_cfg5.sh r04_g_cfg5_b256 > gpurun_out/r04_g_cfg5_b256.out 2>&1; tail -1 gpurun_out/r04_g_cfg5_b256.out
for g in 1 2 4 8; do
  FK_NT_GRID_MULT=$g timeout -k 10 200 python tools/occupant_probe.py 48 2>&1 | grep -v "^$" | tee -a gpurun_out/r04_g_occupant_probe.txt
done
for rep in 1 2; do for g in 1 2 4 8; do
  FK_NT_GRID_MULT=$g timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity --no-other-configs 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid mult $g', 'ms/step', d['ms_per_step'])" | tee -a gpurun_out/r04_g_grid_mult_step.txt
done; done
