#!/bin/bash
# rocprofv3 kernel trace of the benchmark step (BENCH_ARGS="--head ce": the CE-head variant) -> profiles-style summary.  usage: tools/profile_step.sh OUT_PREFIX "title" (run from the repo root on the GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-prof_step}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d ${OUT}_db -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-timers --no-parity --no-other-configs ${BENCH_ARGS} > ${OUT}.log 2>&1
cd $ROOT
python3 tools/prof_summary.py $(find ${OUT}_db -name "*.db" | head -1) 6 ${OUT} "${2:-kernel trace}: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-timers --no-parity --no-other-configs ${BENCH_ARGS} (6 steps in the trace)"
rm -rf ${OUT}_db
tail -1 ${OUT}.log | cut -c1-300
