#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp_backward_fused" 2>&1 | tail -5
for lib in base mf_NODMA mf_NOST mf_NOLD mf_ALL; do
  if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
  timeout -k 10 100 python tools/mlp_fused_bench.py | tee -a gpurun_out/r04_p_mlp_fused_ablate.txt
done
