#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "token_on_the_lane or gemm or ring" 2>&1 | tail -8
for rep in 1 2 3; do for f in 0 1; do
  echo "== FK_NT384_FUSED=$f" | tee -a gpurun_out/r04_z_nt384_fused.txt
  FK_NT384_FUSED=$f timeout -k 10 200 python tools/gemm_bench.py 5 | grep -E "proj\+res" | tee -a gpurun_out/r04_z_nt384_fused.txt
  FK_NT384_FUSED=$f timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity --no-other-configs 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FK_NT384_FUSED=$f', 'ms/step', d['ms_per_step'])" | tee -a gpurun_out/r04_z_nt384_fused.txt
done; done
