#!/bin/bash
# Round-4 call 2: the suite on the advice fixes; non-temporal epilogue stores (variant ntst) against the in-tree library: GEMM bench, TCC counters, step.
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_b_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r04_b_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_b_gpu_tests.log
V=$PWD/frankenstein_amd/variants/lib_ntst.so
for rep in 1 2; do
  echo "== base" | tee -a gpurun_out/r04_b_ntst_gemm.txt
  timeout -k 10 200 python tools/gemm_bench.py 5 | tee -a gpurun_out/r04_b_ntst_gemm.txt
  echo "== ntst" | tee -a gpurun_out/r04_b_ntst_gemm.txt
  FRANKEN_HIP_LIB=$V timeout -k 10 200 python tools/gemm_bench.py 5 | tee -a gpurun_out/r04_b_ntst_gemm.txt
done
for rep in 1 2 3; do
  for lib in base ntst; do
    if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$V; fi
    timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms/step', d['ms_per_step'])" | tee -a gpurun_out/r04_b_ntst_step.txt
  done
done
export FRANKEN_HIP_LIB=$V
timeout -k 10 400 bash tools/pmc_tcc.sh r04_b_pmc_tcc_ntst > /dev/null 2>&1 || echo "pmc_tcc failed"
