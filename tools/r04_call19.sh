#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r04_s_gpu_tests_fused_default.txt
FK_MLP_BWD_FUSED=0 timeout -k 10 700 python -m pytest tests -m gpu -q 2>&1 | tail -1 | tee -a gpurun_out/r04_s_gpu_tests_fused_default.txt
timeout -k 10 300 python tools/determinism_probe.py 2>&1 | tail -5 | tee gpurun_out/r04_s_determinism.txt
