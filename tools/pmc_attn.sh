#!/bin/bash
# SQ instruction-mix / wait counters of the attention kernels: three rocprofv3 --pmc passes over tools/attn_bench.py (counters only, no tracing),
# summed by tools/pmc_sq.py.  usage: tools/pmc_attn.sh OUT_PREFIX   (run from the repo root on the GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_attn}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d ${OUT}_a -- python3 $ROOT/tools/attn_bench.py 1 > ${OUT}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES -d ${OUT}_b -- python3 $ROOT/tools/attn_bench.py 1 > ${OUT}_b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE -d ${OUT}_c -- python3 $ROOT/tools/attn_bench.py 1 > ${OUT}_c.log 2>&1
cd $ROOT
python3 tools/pmc_sq.py $(find ${OUT}_a ${OUT}_b ${OUT}_c -name "*.db") > ${OUT}.txt
rm -rf ${OUT}_a ${OUT}_b ${OUT}_c
cat ${OUT}.txt
