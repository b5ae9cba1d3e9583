#!/bin/bash
# tools/build_gemm_variant.sh NAME [extra hipcc flags...]: frankenstein_amd/variants/lib_NAME.so = the current objects with gemm.hip
# recompiled with the extra flags (e.g. -DFK_RING_PROBE_NOMMA); select it at run time with FRANKEN_HIP_LIB.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/frankenstein_amd/variants /tmp/fkvar
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form \
  -I$ROOT/frankenstein_amd/csrc -I$ROOT/include "$@" -c -x hip $ROOT/frankenstein_amd/csrc/gemm.hip -o /tmp/fkvar/gemm_$NAME.o
B=$ROOT/frankenstein_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/frankenstein_amd/variants/lib_$NAME.so /tmp/fkvar/gemm_$NAME.o $B/attention.hip.o $B/norm.hip.o $B/elementwise.hip.o $B/loss_optim.hip.o $B/pipeline.hip.o $B/conv.hip.o $B/decode.hip.o
echo built $ROOT/frankenstein_amd/variants/lib_$NAME.so
