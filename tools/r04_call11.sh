#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_k_gpu_tests_under_knobs.txt
: > $O
run() { echo "== $*" | tee -a $O; env "$@" timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -1 | tee -a $O; }
run FK_DUMMY=1
run FK_NT_GRID_MULT=1
run FK_NT_GRID_MULT=8
run FK_ATTN_NO_IDENT_PRESCALE=1
run FK_TN_SPLIT_MULT=2
run FK_TEST_POISON=0
