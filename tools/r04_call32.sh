#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_mlp_size_sweep.txt
: > $O
FK_MLP_UP_FUSED=1 FK_QKV_FUSED=1 timeout -k 10 200 python tools/mlp_size_sweep.py | tee -a $O
FK_MLP_UP_FUSED=0 FK_QKV_FUSED=0 timeout -k 10 200 python tools/mlp_size_sweep.py | tee -a $O
