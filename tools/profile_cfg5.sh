#!/bin/bash
# rocprofv3 kernel trace of the SimpleMAE (BASELINE configs[4]) training step at per-GPU batch B (default 32): which kernels fill the step.
# usage: B=32 tools/profile_cfg5.sh OUT_PREFIX   (run from the repo root on the GPU box) -> gpurun_out/OUT_PREFIX.md / .csv
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-cfg5}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d ${OUT}_db -- python3 $ROOT/tools/small_cfg_profile.py > ${OUT}.log 2>&1
cd $ROOT
python3 tools/prof_summary.py $(find ${OUT}_db -name "*.db" | head -1) 20 ${OUT} "SimpleMAE (cfg5) training step, B = ${B:-32}: rocprofv3 --kernel-trace --stats -- python3 tools/small_cfg_profile.py (20 steps in the trace)"
rm -rf ${OUT}_db
