#!/bin/bash
# generated MLP-backward step: one LDS wait per feature tile against one per MFMA
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_mf_wait_per_tile.txt
: > $O
timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp_backward_fused" 2>&1 | tail -2 | tee -a $O
V=$PWD/frankenstein_amd/variants/lib_mf_w0.so
B="--steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity --no-other-configs"
for rep in 1 2 3; do
  timeout -k 10 100 python tools/mlp_fused_bench.py | tee -a $O
  FRANKEN_HIP_LIB=$V timeout -k 10 100 python tools/mlp_fused_bench.py | tee -a $O
  timeout -k 10 120 python bench.py $B 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one wait per tile   ms/step', d['ms_per_step'])" | tee -a $O
  FRANKEN_HIP_LIB=$V timeout -k 10 120 python bench.py $B 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one wait per MFMA   ms/step', d['ms_per_step'])" | tee -a $O
done
