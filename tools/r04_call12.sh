#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do for lib in base ntl1 ntl2; do
  if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
  echo "== $lib" | tee -a gpurun_out/r04_l_nt_loads.txt
  timeout -k 10 200 python tools/gemm_bench.py 5 | grep -E "dswiglu|proj\+res|down\+res" | tee -a gpurun_out/r04_l_nt_loads.txt
  timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity --no-other-configs 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms/step', d['ms_per_step'])" | tee -a gpurun_out/r04_l_nt_loads.txt
done; done
