#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
for lib in base mf_NODMA mf_NOMATH mf_NOST mf_NOLD mf_ALL; do
  if [ $lib = base ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$PWD/frankenstein_amd/variants/lib_$lib.so; fi
  timeout -k 10 100 python tools/mlp_fused_bench.py | tee -a gpurun_out/r04_n_mlp_fused_ablate.txt
done
