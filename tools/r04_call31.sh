#!/bin/bash
# SimpleMAE B=256 (encoder M = 38 400 rows, decoder 153 600): the token-on-the-lane MLP kernels against the tiled ones at these sizes
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_cfg5_mlp_routing.txt
: > $O
for rep in 1 2; do
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  echo "== FK_MLP_UP_FUSED=$1 FK_MLP_BWD_FUSED=$2" | tee -a $O
  FK_MLP_UP_FUSED=$1 FK_MLP_BWD_FUSED=$2 timeout -k 10 200 python tools/other_configs_bench.py | grep cfg5 | tee -a $O
done; done
