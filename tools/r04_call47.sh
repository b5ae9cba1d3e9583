#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_gather_vec.txt
: > $O
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -m gpu -x -q -k "gather or mae or MAE" 2>&1 | tail -3 | tee -a $O
for rep in 1 2; do timeout -k 10 200 python tools/other_configs_bench.py | grep cfg5 | tee -a $O; done
