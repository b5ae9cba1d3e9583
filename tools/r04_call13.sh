#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp_backward_fused" 2>&1 | tail -15
timeout -k 10 200 python tools/gemm_bench.py 5 | grep -E "dswiglu|d_up|fused" | tee gpurun_out/r04_m_mlp_fused.txt
