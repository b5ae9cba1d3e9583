#!/bin/bash
# tools/build_variant.sh NAME [attention source] [extra hipcc flags...]: build frankenstein_amd/variants/lib_NAME.so from the current
# objects with attention.hip (or the given file) recompiled with extra flags; select it at run time with FRANKEN_HIP_LIB.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=${2:-$ROOT/frankenstein_amd/csrc/attention.hip}; shift; shift || true
mkdir -p $ROOT/frankenstein_amd/variants /tmp/fkvar
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form \
  -I$ROOT/frankenstein_amd/csrc -I$ROOT/include "$@" -c -x hip $SRC -o /tmp/fkvar/attention_$NAME.o 2>/dev/null
B=$ROOT/frankenstein_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/frankenstein_amd/variants/lib_$NAME.so $B/gemm.hip.o /tmp/fkvar/attention_$NAME.o $B/norm.hip.o $B/elementwise.hip.o $B/loss_optim.hip.o $B/pipeline.hip.o $B/conv.hip.o $B/decode.hip.o
echo built $ROOT/frankenstein_amd/variants/lib_$NAME.so
