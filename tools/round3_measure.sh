#!/bin/bash
# Round-3 measurement bundle (one GPU-box call, run from the repo root): GPU suite (default and with the weight-gradient side stream forced
# on), the benchmark lines (L1 head incl. CPU baseline, CE head), per-kernel traces of both, the counter passes, the attention SQ counters.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
P=${1:-r03_f}
O=$ROOT/gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/${P}_gpu_tests.log 2>&1; tail -2 $O/${P}_gpu_tests.log
FK_WGRAD_STREAM=1 timeout -k 10 900 python -m pytest tests -m gpu -q > $O/${P}_gpu_tests_wgrad_stream.log 2>&1; tail -2 $O/${P}_gpu_tests_wgrad_stream.log
python bench.py > $O/${P}_bench.log 2>&1; tail -1 $O/${P}_bench.log | cut -c1-330
python bench.py --head ce --no-cpu-baseline > $O/${P}_bench_ce.log 2>&1; tail -1 $O/${P}_bench_ce.log | cut -c1-330
bash tools/profile_step.sh ${P} "round 3 final" > $O/${P}_profile.out 2>&1; tail -1 $O/${P}_profile.out | cut -c1-200
BENCH_ARGS="--head ce" bash tools/profile_step.sh ${P}_ce "round 3 final, CE-head variant" > $O/${P}_ce_profile.out 2>&1; tail -1 $O/${P}_ce_profile.out | cut -c1-200
bash tools/pmc_step.sh ${P} > $O/${P}_pmc_step.out 2>&1; tail -3 $O/${P}_pmc_step.out
bash tools/pmc_attn.sh ${P}_pmc_sq_attention > $O/${P}_pmc_attn.out 2>&1; tail -2 $O/${P}_pmc_attn.out
{ echo "== in-tree library"; python tools/gemm_bench.py 5 2>&1 | grep -E "^nt|^tn|sum"; } > $O/${P}_gemm_bench.txt 2>&1; tail -1 $O/${P}_gemm_bench.txt
bash tools/kt_attn.sh main > $O/${P}_kt_attn.txt 2>&1; grep -E "asm|fwd" $O/${P}_kt_attn.txt
