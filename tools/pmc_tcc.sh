#!/bin/bash
# L2 (TCC) view of the cfg2-shaped GEMM launches: hit / miss / request counters and the memory-side read / write request counters, two
# rocprofv3 --pmc passes over tools/gemm_bench.py (counters only), one table per kernel by tools/pmc_sq.py.  Answers whether the read
# amplification FETCH_SIZE reports for the K = 384 ring kernels is real (misses >> compulsory) or a counting artefact.
# usage: tools/pmc_tcc.sh OUT_PREFIX   (run from the repo root on the GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_tcc}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d ${OUT}_a -- python3 $ROOT/tools/gemm_bench.py 1 > ${OUT}_a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d ${OUT}_b -- python3 $ROOT/tools/gemm_bench.py 1 > ${OUT}_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_WRITE_sum TCC_STREAMING_REQ_sum TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum -d ${OUT}_c -- python3 $ROOT/tools/gemm_bench.py 1 > ${OUT}_c.log 2>&1
cd $ROOT
python3 tools/pmc_sq.py $(find ${OUT}_a ${OUT}_b ${OUT}_c -name "*.db") > ${OUT}.txt
rm -rf ${OUT}_a ${OUT}_b ${OUT}_c
cat ${OUT}.txt
