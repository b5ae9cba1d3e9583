#!/bin/bash
# Round-4 call 4: suite on the scratch-free forward (+ nt GEMM stores by default), forward A/B against the previous attention object.
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_d_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r04_d_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_d_gpu_tests.log
for rep in 1 2 3; do
  for lib in new fwdold; do
    if [ $lib = new ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
    echo "== $lib" | tee -a gpurun_out/r04_d_fwd_ab.txt
    timeout -k 10 120 python tools/attn_bench.py 7 | tee -a gpurun_out/r04_d_fwd_ab.txt
    timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'ms/step', d['ms_per_step'])" | tee -a gpurun_out/r04_d_fwd_ab.txt
  done
done
unset FRANKEN_HIP_LIB
timeout -k 10 300 python bench.py > gpurun_out/r04_d_bench.log 2>&1 && tail -1 gpurun_out/r04_d_bench.log | cut -c1-300
