#!/bin/bash
# Round-4 call 3: the suite (guard bands on by default, B = 32 / B = 8 fixtures, advice fixes), then non-temporal store variants in alternating runs.
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_c_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r04_c_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_c_gpu_tests.log
for rep in 1 2 3; do
  for lib in base g1 g2 a1 n1 g1a1 g2a1 g1a1n1; do
    if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
    timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms/step', d['ms_per_step'])" | tee -a gpurun_out/r04_c_nt_step.txt
  done
done
for rep in 1 2; do
  for lib in base a1; do
    if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
    echo "== $lib" | tee -a gpurun_out/r04_c_nt_attn.txt
    timeout -k 10 120 python tools/attn_bench.py 5 | tee -a gpurun_out/r04_c_nt_attn.txt
  done
done
