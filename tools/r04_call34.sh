#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_fill_routing_ab.txt
: > $O
for rep in 1 2 3; do
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  echo "== FK_MLP_UP_FUSED=$1 FK_QKV_FUSED=$2" | tee -a $O
  FK_MLP_UP_FUSED=$1 FK_QKV_FUSED=$2 timeout -k 10 200 python tools/other_configs_bench.py | grep "cfg5.*256" | tee -a $O
done; done
