#!/bin/bash
# rocprofv3 counter passes over one benchmark step (counters only, --kernel-trace, no other tracing): HBM bytes per kernel (FETCH_SIZE and
# WRITE_SIZE in separate passes) and MFMA-pipe busy cycles.  usage: tools/pmc_step.sh PREFIX   -> gpurun_out/PREFIX_pmc_traffic.json, _pmc_mfma_util.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r02}
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-timers --no-parity --no-other-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${OUT}_f -- $B > ${OUT}_pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${OUT}_w -- $B > ${OUT}_pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU -d ${OUT}_m -- $B > ${OUT}_pmc_m.log 2>&1
cd $ROOT
python3 tools/pmc_traffic.py $(find ${OUT}_f -name "*.db" | head -1) $(find ${OUT}_w -name "*.db" | head -1) ${OUT}_pmc_traffic.json > ${OUT}_pmc_traffic.txt; head -8 ${OUT}_pmc_traffic.txt
python3 tools/pmc_mfma_util.py $(find ${OUT}_m -name "*.db" | head -1) 2 ${OUT}_pmc_mfma_util.json > ${OUT}_pmc_mfma_util.txt; head -14 ${OUT}_pmc_mfma_util.txt
rm -rf ${OUT}_f ${OUT}_w ${OUT}_m
