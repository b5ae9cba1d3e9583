#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_f_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r04_f_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04_f_gpu_tests.log
for rep in 1 2; do
echo "== identity prescale (default)" | tee -a gpurun_out/r04_f_other_configs.txt
timeout -k 10 300 python tools/other_configs_bench.py 2>&1 | grep "cfg" | tee -a gpurun_out/r04_f_other_configs.txt
echo "== FK_ATTN_NO_IDENT_PRESCALE=1" | tee -a gpurun_out/r04_f_other_configs.txt
FK_ATTN_NO_IDENT_PRESCALE=1 timeout -k 10 300 python tools/other_configs_bench.py 2>&1 | grep "cfg" | tee -a gpurun_out/r04_f_other_configs.txt
done
