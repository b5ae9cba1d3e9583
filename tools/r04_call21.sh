#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
for lib in base mf_spreadst; do
  if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
  timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp_backward_fused" 2>&1 | tail -1
done
for rep in 1 2 3; do for lib in base mf_spreadst; do
  if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
  timeout -k 10 100 python tools/mlp_fused_bench.py | tee -a gpurun_out/r04_v_mlp_fused_variants.txt
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity --no-other-configs 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms/step', d['ms_per_step'])" | tee -a gpurun_out/r04_v_mlp_fused_variants.txt
done; done
