#!/bin/bash
# generated-stream fused MLP backward: stamps, issue-budget variants, in-step A/B against the hipcc kernel (FK_MLP_BWD_ASM=0)
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_mf_asm_step_ab.txt
: > $O
FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_mf_stamp.so timeout -k 10 120 python tools/stamp_mlp.py > gpurun_out/r04_mf_stamps_asm.txt 2>&1; sed -n 1,13p gpurun_out/r04_mf_stamps_asm.txt
B="--steps 10 --warmup 3 --no-cpu-baseline --no-timers --no-parity --no-other-configs"
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 120 python bench.py $B 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label  ms/step', d['ms_per_step'])" | tee -a $O
}
for rep in 1 2 3; do
  run "stream budget 7 (in-tree)" X=1
  run "hipcc kernel (FK_MLP_BWD_ASM=0)" FK_MLP_BWD_ASM=0
  run "stream budget 6" FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_mf_b6.so
  run "stream budget 8" FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_mf_b8.so
  run "stream budget 9" FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_mf_b9.so
done
