#!/bin/bash
# Everything the round's profiles/ are made from, in one GPU-box call (run from the repo root): full GPU suite, the benchmark line, the
# per-kernel trace of the step, the counter passes, the attention SQ counters, the clock probe, the issue probes and the GEMM phase split.
# Each step writes under gpurun_out/; a failing step stops the rest (set -e).  usage: tools/round_end_measure.sh PREFIX (e.g. r02_f)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
P=${1:-r02_f}
O=$ROOT/gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${P}_gpu_tests.log 2>&1; tail -2 $O/${P}_gpu_tests.log
python bench.py > $O/${P}_bench.log 2>&1; tail -1 $O/${P}_bench.log | cut -c1-330
bash tools/profile_step.sh ${P} "round 2 final" > $O/${P}_profile.out 2>&1; tail -1 $O/${P}_profile.out | cut -c1-200
bash tools/pmc_step.sh ${P} > $O/${P}_pmc_step.out 2>&1; tail -3 $O/${P}_pmc_step.out
bash tools/pmc_attn.sh ${P}_pmc_sq_attention > $O/${P}_pmc_attn.out 2>&1; tail -2 $O/${P}_pmc_attn.out
FRANKEN_HIP_LIB=$ROOT/frankenstein_amd/variants/lib_stamp.so timeout -k 5 100 python tools/clock_attn.py > $O/${P}_clock_attn.txt 2>&1; tail -2 $O/${P}_clock_attn.txt
for p in valu_rate mfma_mix mfma_lds mfma_shape mfma16_layout dma_rowwidth; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probes/$p.hip -o /tmp/$p && timeout -k 5 100 /tmp/$p > $O/${P}_probe_$p.txt 2>&1; done; tail -2 $O/${P}_probe_mfma_lds.txt
{ echo "== in-tree library"; python tools/gemm_bench.py 5 2>&1 | grep -E "^nt|^tn|sum";
  for v in noepi nomma; do echo "== $v (probe build: $( [ $v = noepi ] && echo 'main loops only, no epilogue' || echo 'fetch stream and epilogue only, no MFMA' ))";
    FRANKEN_HIP_LIB=$ROOT/frankenstein_amd/variants/lib_$v.so python tools/gemm_bench.py 3 2>&1 | grep -E "^nt|sum"; done; } > $O/${P}_gemm_phase_split.txt 2>&1
tail -3 $O/${P}_gemm_phase_split.txt
bash tools/kt_attn.sh main > $O/${P}_kt_attn.txt 2>&1; grep -E "asm|fwd" $O/${P}_kt_attn.txt
