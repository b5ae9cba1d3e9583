#!/bin/bash
# tools/pmc_attn.sh for a library variant: tools/pmc_attn_variant.sh VARIANT OUT_PREFIX
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export FRANKEN_HIP_LIB=$ROOT/frankenstein_amd/variants/lib_$1.so
exec $ROOT/tools/pmc_attn.sh $2
