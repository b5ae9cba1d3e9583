"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports exactly the
symbols include/franken_hip.h declares (no compute calls: there is no GPU here)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    txt = (ROOT / "include" / "franken_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fk_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    from frankenstein_amd import build
    return build.build(verbose=False)


def test_header_and_binding_agree(built):
    from frankenstein_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 30
    assert sorted(_lib.SIGNATURES) == syms, set(_lib.SIGNATURES) ^ set(syms)


def test_library_loads_and_exports_every_symbol(built):
    import ctypes
    from frankenstein_amd import _lib
    h = ctypes.CDLL(str(built))
    for s in declared_symbols():
        assert hasattr(h, s), f"{s} declared in include/franken_hip.h but not exported"
    lib = _lib.lib()
    assert lib.fk_version() == 302
    assert lib.fk_last_error() is not None


def test_graft_entry_build_runs(built):
    """the driver's build check (`__graft_entry__.build()`): builds, loads, and finds the version the header declares"""
    import __graft_entry__ as g
    g.build()
    from frankenstein_amd import _lib
    txt = (ROOT / "include" / "franken_hip.h").read_text()
    assert _lib.lib().fk_version() == int(re.search(r"#define FK_VERSION (\d+)", txt).group(1)) == 302


def test_argument_validation_needs_no_gpu(built):
    """Bad arguments are rejected on the host before any launch (error convention of SURVEY §8b)."""
    from frankenstein_amd import _lib
    lib = _lib.lib()
    rc = lib.fk_gemm_nt(None, 0, None, 0, None, 0, 0, 0, 0, None, None, 0, 0, 7, 0, None)
    assert rc == -1 and b"dtype" in lib.fk_last_error()
    rc = lib.fk_gemm_nt(16, 12, 16, 12, 16, 8, 8, 8, 12, None, None, 0, 0, _lib.FK_BF16, _lib.FK_BF16, None)
    assert rc == -1 and b"multiples" in lib.fk_last_error()
    rc = lib.fk_attn_fwd(16, 16, 16, 16, None, 1, 1, 8, 8, 24, 0, 24, 0, 24, 0, 24, 0, 24, 0, 0, 0, 0, None, None, 1.0, 0, 0, None)
    assert rc == -1 and b"head_dim" in lib.fk_last_error()
    # FK_ATTN_Q_PRESCALED exists for bf16 / head_dim 64 only
    rc = lib.fk_attn_fwd(16, 16, 16, 16, None, 1, 1, 8, 8, 32, 0, 32, 0, 32, 0, 32, 0, 32, 0, 0, 0, 0, None, None, 1.0,
                         _lib.ATTN_Q_PRESCALED, _lib.FK_BF16, None)
    assert rc == -1 and b"FK_ATTN_Q_PRESCALED" in lib.fk_last_error()
    rc = lib.fk_patchify(16, 16, 1, 10, 4, 3, 8, 0, None)
    assert rc == -1
    # fk_sample_topk: an output buffer without its width is refused (the device-side step counter is not trusted to stay inside it)
    rc = lib.fk_sample_topk(16, 8, 1, 8, 1.0, 0, 16, 16, None, 16, 16, 4, 0, 16, None)
    assert rc == -1 and b"out_cols" in lib.fk_last_error()


def test_integration_snippet_matches_the_binding():
    """The ctypes example a maintainer would copy from INTEGRATION.md: its argtypes equal _lib.SIGNATURES["fk_attn_fwd"] and the call
    passes exactly that many arguments (the round-1 version of the snippet had drifted from the header)."""
    import ctypes
    from frankenstein_amd import _lib
    txt = (ROOT / "INTEGRATION.md").read_text()
    snippet = txt[txt.index("```python\n# models/brainformer.py"):]
    snippet = snippet[: snippet.index("```", 10)]
    line = next(l for l in snippet.splitlines() if l.startswith("_fk.fk_attn_fwd.argtypes"))
    env = {"ctypes": ctypes, "I64": ctypes.c_int64, "P": ctypes.c_void_p, "INT": ctypes.c_int, "F32": ctypes.c_float}
    argtypes = eval(line.split("=", 1)[1], env)
    assert argtypes == _lib.SIGNATURES["fk_attn_fwd"][1]
    call = snippet[snippet.index("_fk.fk_attn_fwd(q.data_ptr()"):]
    call = call[call.index("(") + 1: call.index("if rc:")]
    call = re.sub(r"#.*", "", call)
    depth, nargs, cur = 0, 0, ""
    for ch in call:
        if ch in "([":
            depth += 1
        if ch in ")]":
            if depth == 0:
                break
            depth -= 1
        if ch == "," and depth == 0:
            nargs += 1
            cur = ""
        else:
            cur += ch
    nargs += 1 if cur.strip() else 0
    assert nargs == len(argtypes), (nargs, len(argtypes))
    # and the header's prototype has the same number of parameters
    hdr = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "franken_hip.h").read_text(), flags=re.S)
    proto = re.search(r"int fk_attn_fwd\((.*?)\);", hdr, flags=re.S).group(1)
    assert len(proto.split(",")) == len(argtypes)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from frankenstein_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.FrankenHipError, match="no CPU / PyTorch fallback"):
        _lib.lib()
