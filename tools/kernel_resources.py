"""What hipcc made of the hand-scheduled kernels: register / scratch / occupancy figures from `-Rpass-analysis=kernel-resource-usage`
and, from the device assembly, whether any spill traffic sits inside an MFMA loop.

The generated instruction streams (csrc/attn_*_asm.inc) hard-code their temporaries as clobbers and leave the accumulators, fragments and
addresses to hipcc; a compiler update that spills inside the loop or pushes a kernel past 256 registers (one wave per SIMD instead of
two) would be silent until a performance run.  tests/test_codegen_cpu.py holds the compiled kernels to the documented bounds.

    python tools/kernel_resources.py            # table for the hand-scheduled kernels (all nine sources are compiled)
"""
from __future__ import annotations

import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def compile_isa(src: str, outdir: Path):
    """device-only assembly + resource remarks of csrc/<src> with the build's own flags -> (path of the .s, remark text)"""
    from frankenstein_amd import build as B
    out = Path(outdir) / (src + ".s")
    cmd = [B.HIPCC, *B.flags_for(src), "-S", "--cuda-device-only", f"-I{ROOT / 'include'}", "-Rpass-analysis=kernel-resource-usage",
           "-o", str(out), str(B.CSRC / src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return out, r.stderr


def parse_remarks(text: str) -> dict:
    """mangled kernel name -> {'VGPRs': int, 'AGPRs': int, 'ScratchSize': bytes per lane, 'Occupancy': waves per SIMD,
    'VGPRs Spill': int, 'SGPRs Spill': int, 'TotalSGPRs': int}"""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def mfma_loops(asm_path: Path, kernel: str):
    """[(first line, last line, #MFMA, #scratch loads/stores)] for every loop (backward branch) of `kernel` (mangled name) that
    contains matrix instructions"""
    lines = Path(asm_path).read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i + 1) < i:
            seg = body[labels[m.group(1)]:i]
            nm = sum("v_mfma" in s for s in seg)
            if nm:
                loops.append((labels[m.group(1)], i, nm, sum(bool(re.match(r"\s+scratch_(load|store)", s)) for s in seg)))
    return loops


def hot_loop(loops):
    """the innermost loop (no other MFMA loop nested in it) with the most matrix instructions: (first, last, #MFMA, #scratch) or None"""
    inner = [l for l in loops if not any(o is not l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    return max(inner, key=lambda l: l[2]) if inner else None


def survey(outdir: Path, sources=None):
    """every source of the library by default (build.SOURCES): the packed-fp32 guard has to see elementwise.hip (fk_rope, one of the
    recorded victims of DESIGN.md 5.4) and the rest, not only the two files with hand-scheduled kernels"""
    if sources is None:
        from frankenstein_amd import build as B
        sources = tuple(B.SOURCES)
    outdir.mkdir(parents=True, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, len(sources))) as ex:
        res = list(ex.map(lambda s: compile_isa(s, outdir), sources))
    table = {}
    for (asm, remarks) in res:
        for name, r in parse_remarks(remarks).items():
            r["_asm"] = asm
            table[name] = r
    return table


if __name__ == "__main__":
    import tempfile
    tab = survey(Path(tempfile.mkdtemp()))
    for name, r in sorted(tab.items()):
        if any(s in name for s in ("asm_kernel", "ring", "tn_big", "_ps_kernel")):
            loops = mfma_loops(r["_asm"], name)
            inloop = max((l[3] for l in loops), default=0)
            hot = hot_loop(loops)
            print(f"{name[:84]:84s} VGPR {r['VGPRs']:3d} scratch {r['ScratchSize']:3d} B/lane spills {r['VGPRs Spill']:3d} "
                  f"waves/SIMD {r['Occupancy']} SGPR {r['TotalSGPRs']:3d} scratch ops: any MFMA loop {inloop}, "
                  f"hot loop ({hot[2] if hot else 0} MFMAs) {hot[3] if hot else 0}")
