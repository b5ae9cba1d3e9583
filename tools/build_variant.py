"""tools/build_variant.py NAME SRC[,SRC...] [extra hipcc flags...]: frankenstein_amd/variants/lib_NAME.so = the in-tree objects with the named
sources (e.g. gemm.hip) recompiled with the build's own flags (frankenstein_amd/build.py) plus the extra ones; select it at run time
with FRANKEN_HIP_LIB for A/B timing inside one gpurun call."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from frankenstein_amd import build as B

name, srcs, extra = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build(verbose=False)
out = ROOT / "frankenstein_amd" / "variants"
tmp = Path("/tmp/fkvar") / name
out.mkdir(exist_ok=True)
tmp.mkdir(parents=True, exist_ok=True)


def cc(s):
    o = tmp / (s + ".o")
    r = subprocess.run([B.HIPCC, *B.flags_for(s), *extra, "-c", str(B.CSRC / s), "-o", str(o)], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr[-3000:])
    return o


with ThreadPoolExecutor(max_workers=4) as ex:
    new = dict(zip(srcs, ex.map(cc, srcs)))
objs = [str(new.get(s, B.CSRC / "build" / (s + ".o"))) for s in B.SOURCES]
lib = out / f"lib_{name}.so"
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *objs], check=True)
print("built", lib)
