#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_models_gpu.py -m gpu -x -q -k "cfg5 or simple_mae" 2>&1 | tail -4
for rep in 1 2; do for lib in base norope; do
  if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
  echo "== $lib" | tee -a gpurun_out/r04_j_rope_table_probe.txt
  timeout -k 10 200 python tools/gemm_bench.py 5 | grep -E "qkv|swiglu " | tee -a gpurun_out/r04_j_rope_table_probe.txt
done; done
