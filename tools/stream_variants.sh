#!/bin/bash
# tools/stream_variants.sh GEN NAME=ENVVAR[,ENVVAR...] ...: timing variants of one generated stream (results WRONG, timing only): for every
# NAME the generator tools/gen/gen_GEN_asm.py is run with the given FK_GEN_* variables set to 1 (or VAR:VALUE), attention.hip is compiled
# against that copy of attn_GEN_asm.inc and linked as frankenstein_amd/variants/lib_NAME.so.  Example:
#   tools/stream_variants.sh dkdvw w_nolgkm=FK_GEN_ABLATE_LGKM w_novalu=FK_GEN_ABLATE_VALU w_v8=FK_GEN_VALU_UNITS:8
# FK_VARIANT_FLAGS: extra hipcc flags for every variant (e.g. -DFK_FWD_PROBE_NOREDO for ablated forward streams).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
GEN=$1; shift
B=$ROOT/frankenstein_amd/csrc/build
mkdir -p $ROOT/frankenstein_amd/variants
for spec in "$@"; do
  (
  NAME=${spec%%=*}; VARS=${spec#*=}
  D=/tmp/fkvar/src_$NAME; mkdir -p $D
  cp $ROOT/frankenstein_amd/csrc/attention.hip $D/
  env_args=""
  IFS=',' read -ra VS <<< "$VARS"
  for v in "${VS[@]}"; do if [[ "$v" == *:* ]]; then env_args="$env_args ${v%%:*}=${v#*:}"; elif [ -n "$v" ]; then env_args="$env_args $v=1"; fi; done
  env $env_args python3 $ROOT/tools/gen/gen_${GEN}_asm.py $D/attn_${GEN}_asm.inc > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form \
    -I$ROOT/frankenstein_amd/csrc -I$ROOT/include $FK_VARIANT_FLAGS -c $D/attention.hip -o $D/attention.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/frankenstein_amd/variants/lib_$NAME.so $B/gemm.hip.o $D/attention.o $B/norm.hip.o $B/elementwise.hip.o $B/loss_optim.hip.o $B/pipeline.hip.o $B/conv.hip.o $B/decode.hip.o $B/head_ce.hip.o
  echo built lib_$NAME.so
  ) &
done
wait
