#!/bin/bash
# Round-4 measurement bundle: suite, bench lines (L1 incl. CPU baseline / parity / other configs; CE head), kernel trace, counter passes, GEMM / attention benches, TN occupant probe
set -o pipefail
P=${1:-r04_i}
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/${P}_gpu_tests.log 2>&1; tail -2 $O/${P}_gpu_tests.log
python bench.py > $O/${P}_bench.log 2>&1; tail -1 $O/${P}_bench.log | cut -c1-330
python bench.py --head ce --no-cpu-baseline --no-other-configs > $O/${P}_bench_ce.log 2>&1; tail -1 $O/${P}_bench_ce.log | cut -c1-200
bash tools/profile_step.sh ${P} "round 4" > $O/${P}_profile.out 2>&1; tail -1 $O/${P}_profile.out | cut -c1-200
bash tools/pmc_step.sh ${P} > $O/${P}_pmc_step.out 2>&1; tail -3 $O/${P}_pmc_step.out
{ echo "== in-tree library"; python tools/gemm_bench.py 5 2>&1 | grep -E "^nt|^tn|sum"; } > $O/${P}_gemm_bench.txt 2>&1; tail -1 $O/${P}_gemm_bench.txt
python tools/attn_bench.py 7 > $O/${P}_attn_bench.txt 2>&1; cat $O/${P}_attn_bench.txt
for m in 1 2 4; do FK_TN_SPLIT_MULT=$m timeout -k 10 200 python tools/occupant_probe.py 48 2>&1 | grep -E "^FK|tn dW" | sed "s/^/TN_SPLIT_MULT=$m  /" | tee -a $O/${P}_occupant_tn.txt; done
