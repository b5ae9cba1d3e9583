#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "mlp_backward_fused or mlp_branch_with_the_fused or fused_mlp_backward_keeps or determin" 2>&1 | tail -3
for lib in base mf_ALLm; do
  if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
  timeout -k 10 100 python tools/mlp_fused_bench.py | tee -a gpurun_out/r04_t_mlp_fused.txt
done
unset FRANKEN_HIP_LIB
for rep in 1 2 3; do for f in 0 1; do
  FK_MLP_BWD_FUSED=$f timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity --no-other-configs 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FK_MLP_BWD_FUSED=$f', 'ms/step', d['ms_per_step'])" | tee -a gpurun_out/r04_t_mlp_fused_step.txt
done; done
