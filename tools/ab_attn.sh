#!/bin/bash
# A/B of attention library variants inside ONE gpurun call: tools/ab_attn.sh ROUNDS name1 name2 ... ("main" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset FRANKEN_HIP_LIB; else export FRANKEN_HIP_LIB=$ROOT/frankenstein_amd/variants/lib_$v.so; fi
  echo "== $v (rep $rep)"; python3 $ROOT/tools/attn_bench.py $R 2>&1 | grep "attn_"
done; done
