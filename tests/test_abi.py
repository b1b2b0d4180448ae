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


def test_launch_list_records_and_replays_c_abi_calls():
    """ops.LaunchList mechanics without a GPU: calls made through _lib.lib() while recording are kept in order with their
    arguments, replay re-issues them and surfaces a failing return code as LlarkHipError; recording does not nest."""
    import pytest
    from llark_amd import _lib as LB
    from llark_amd import ops as O
    O._stream = lambda: 0                                     # no device here; restored below
    try:
        with O.LaunchList.record() as ll:
            rc = LB.lib().llark_prior_attn(None, 0, 1, 64, 48, 2, 8, 1, None, None, 0, None)     # rejected before any HIP call
            assert rc == -1
            with pytest.raises(RuntimeError, match="does not nest"):
                O.LaunchList.record().__enter__()
        assert [c[1] for c in ll.calls] == ["llark_prior_attn"] and ll.calls[0][2][2:5] == (1, 64, 48)
        assert LB._recorder is None and not isinstance(LB.lib(), LB._RecordingProxy)
        with pytest.raises(LB.LlarkHipError, match="llark_prior_attn"):
            ll.replay()
    finally:
        import torch
        O._stream = lambda: torch.cuda.current_stream().cuda_stream
