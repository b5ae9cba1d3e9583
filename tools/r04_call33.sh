#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_fill_routing.txt
: > $O
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "token_on_the_lane or swiglu or rope or keypad" 2>&1 | tail -4 | tee -a $O
for rep in 1 2; do
  timeout -k 10 200 python tools/other_configs_bench.py | grep cfg5 | tee -a $O
done
timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity --no-other-configs 2>&1 | tail -1 | cut -c1-300 | tee -a $O
