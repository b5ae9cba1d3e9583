#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_mf_asm_ab.txt
: > $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp_backward_fused" 2>&1 | tail -15 | tee -a $O
for rep in 1 2 3; do
  timeout -k 10 100 python tools/mlp_fused_bench.py | tee -a $O
  FK_MLP_BWD_ASM=0 timeout -k 10 100 python tools/mlp_fused_bench.py | sed 's/in-tree/asm off/' | tee -a $O
done
