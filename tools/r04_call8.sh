#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/graph_cfg2_probe.py 2>&1 | grep "ms/step" | tee gpurun_out/r04_h_graph_cfg2.txt
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ring or gemm" 2>&1 | tail -2
