#!/bin/bash
# tools/build_full_variant.sh NAME [extra hipcc flags...]: build frankenstein_amd/variants/lib_NAME.so with EVERY source recompiled with the
# build's flags plus the extra ones; select it at run time with FRANKEN_HIP_LIB.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/frankenstein_amd/variants /tmp/fkvar/$NAME
OBJS=""
for s in gemm attention norm elementwise loss_optim pipeline conv decode head_ce; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form \
    -I$ROOT/frankenstein_amd/csrc -I$ROOT/include "$@" -c $ROOT/frankenstein_amd/csrc/$s.hip -o /tmp/fkvar/$NAME/$s.o 2>/dev/null &
  OBJS="$OBJS /tmp/fkvar/$NAME/$s.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/frankenstein_amd/variants/lib_$NAME.so $OBJS
echo built $ROOT/frankenstein_amd/variants/lib_$NAME.so
