"""CPU: the C-ABI library builds, loads and exports every symbol include/llark_hip.h declares
(no compute calls without a GPU)."""
import os
import re

from llark_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in_header():
    text = open(os.path.join(ROOT, "include", "llark_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(llark_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    L = _lib.lib()
    names = _declared_in_header()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/llark_hip.h but not exported by libllark_hip.so"


def test_python_signatures_cover_header():
    assert _declared_in_header() == _lib.declared_symbols()


def test_version_and_error_string():
    L = _lib.lib()
    assert L.llark_version() >= 100
    assert isinstance(L.llark_last_error(), bytes)


def test_invalid_arguments_are_reported_not_crashed():
    # argument validation happens before any HIP call, so this is safe without a GPU
    L = _lib.lib()
    rc = L.llark_gemm16(0, 1, 0, None, None, 0, None, 0, None, 0, 0, 0, None, 0, None, 0, None, None, 0, None)
    assert rc == -1
    assert b"gemm16" in L.llark_last_error()
    rc = L.llark_prior_attn(None, 0, 1, 64, 48, 2, 8, 1, None, None, 0, None)
    assert rc == -1


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under llark_amd/ may import, call or link it (only tests/,
    __graft_entry__.smoke() and bench.py's cpu_baseline leg do)."""
    import pathlib
    import re
    root = pathlib.Path(__file__).resolve().parents[1] / "llark_amd"
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|oracle/_build|libjukebox_ref", re.M)
    offenders = [str(p) for p in root.rglob("*.py") if pat.search(p.read_text())]
    inc = re.compile(r"#\s*include\s*[<\"][^>\"]*(oracle|jukebox_ref)[^>\"]*[>\"]")
    offenders += [str(p) for ext in ("*.hip", "*.h") for p in root.rglob(ext) if inc.search(p.read_text())]
    mk = root / "csrc" / "Makefile"
    assert "oracle" not in mk.read_text()
    assert not offenders, offenders
