#!/bin/bash
# keypad mask free tiles: tests, the probe and the other configurations, A/B against the -DFK_KEYPAD_NO_FREE=1 build
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_keypad_free.txt
: > $O
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -m gpu -x -q -k "attention or mae or prescaled" 2>&1 | tail -5 | tee -a $O
V=$PWD/frankenstein_amd/variants/lib_keypad_nofree.so
for rep in 1 2; do
  echo "== free tiles (default)" | tee -a $O
  timeout -k 10 120 python tools/keypad_probe.py | tee -a $O
  timeout -k 10 200 python tools/other_configs_bench.py | grep cfg5 | tee -a $O
  echo "== -DFK_KEYPAD_NO_FREE=1" | tee -a $O
  FRANKEN_HIP_LIB=$V timeout -k 10 120 python tools/keypad_probe.py | tee -a $O
  FRANKEN_HIP_LIB=$V timeout -k 10 200 python tools/other_configs_bench.py | grep cfg5 | tee -a $O
done
