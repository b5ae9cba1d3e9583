#!/bin/bash
# Round-4 first GPU call: state of HEAD (suite, bench), per-frame time against the batch (what a MALL-sized chunk could buy), TCC counters.
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_a_gpu_tests.log 2>&1 && tail -2 gpurun_out/r04_a_gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r04_a_bench.log 2>&1 && tail -1 gpurun_out/r04_a_bench.log | cut -c1-400
for rep in 1 2; do
for b in 32 16 8 4; do
  timeout -k 10 120 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-timers 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', d['config']['per_gpu_batch'], 'ms/step', d['ms_per_step'], 'us/sample', round(1e3*d['ms_per_step']/d['config']['per_gpu_batch'],1))" | tee -a gpurun_out/r04_a_batch_sweep.txt
done
done
timeout -k 10 400 bash tools/pmc_tcc.sh r04_a_pmc_tcc > /dev/null 2>&1 || echo "pmc_tcc failed"
tail -5 gpurun_out/r04_a_pmc_tcc_a.log
