"""Build libfranken_hip.so (gfx950 only) in-tree with hipcc.  `python -m frankenstein_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libfranken_hip.so"
SOURCES = ["gemm.hip", "attention.hip", "norm.hip", "elementwise.hip", "loss_optim.hip", "pipeline.hip", "conv.hip", "decode.hip", "head_ce.hip", "mlp_fused.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs (gfx950 has a unified file), which removes the
# v_accvgpr_read/write traffic between the matrix results and the softmax / epilogue VALU code.
# -fno-slp-vectorize: hipcc's SLP vectorizer turns scalar fp32 pairs (the RoPE rotations x0*c - x1*s, x0*s + x1*c) into
# v_pk_mul_f32 -> v_pk_fma_f32 chains with op_sel half-swaps.  On MI355X those chains returned a WRONG low-half result in lanes 48-63
# whenever waves of another kernel (the small bf16 weight-gradient GEMM: ds_read_b64_tr_b16 + v_mfma) shared the SIMD -- in the dQ / dK
# store of fk_attn_bwd, in the QKV projection's RoPE epilogue and in the plain elementwise fk_rope alike (all operands defined, waits
# irrelevant, the quiet result exact: DESIGN.md 5.4, tools/coresidency_sweep.py).  Scalar code is exact beside the same occupant and no
# slower (51.7 ms/step either way).  Hand-written f32x2 arithmetic (no half-swaps) is unaffected and stays.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def flags_for(src: str):
    """mlp_fused.hip keeps its 208 accumulator registers in the AGPR half (one wave per SIMD, 96 + 192 + 16 stationary registers per lane):
    it is compiled WITHOUT -amdgpu-mfma-vgpr-form; everything else with the common flags."""
    if src == "mlp_fused.hip":
        return [f for f in FLAGS if f not in ("-mllvm", "-amdgpu-mfma-vgpr-form")]
    return FLAGS


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    hdrs = [*sorted(CSRC.glob("*.h")), HERE.parent / "include" / "franken_hip.h", *sorted(CSRC.glob("*.inc"))]   # *.inc: generated streams (tools/gen)
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    jobs = []
    for src in SOURCES:
        o = objdir / (src + ".o")
        if force or _stale(o, [CSRC / src, *hdrs]):
            jobs.append((src, o))

    def cc(job):
        src, o = job
        cmd = [HIPCC, *flags_for(src), "-c", str(CSRC / src), "-o", str(o)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        return o

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [objdir / (s + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
