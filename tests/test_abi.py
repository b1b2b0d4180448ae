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


import pytest


@pytest.mark.gpu
def test_persistent_gemms_on_two_streams_use_separate_caller_owned_workspaces():
    """include/llark_hip.h: compute entry points keep no hidden device state -- the persistent GEMM kernels synchronise through
    a workspace the CALLER creates (llark_workspace_create), one per (device, stream).  Two streams run the persistent kernels
    (f16x2 variant 30 and the lo8 form) concurrently, each with its own workspace; results equal the single-stream run bit for
    bit, and the library allocates nothing after the two workspaces exist."""
    import torch

    from llark_amd import ops

    g = torch.Generator().manual_seed(0)
    m, n, k = 8192, 4800, 1216
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(n, k, generator=g) * 0.02).half().cuda()
    hi, lo = ops.split16(a.cuda(), torch.float16, kmult=64)
    wt = torch.zeros((n, hi.shape[1]), dtype=torch.float16, device="cuda")
    wt[:, :k] = w
    lo8 = torch.randint(0, 120, (m, hi.shape[1]), dtype=torch.uint8, device="cuda")
    sw = ops.lo8_weight_exponent(wt)
    w8 = ops.pack_weight_lo8(wt, sw)

    def run(kind, out):
        if kind == 0:
            ops.gemm16(hi, lo, wt, None, n, ops.EPI_F32, c=out, variant=30)
        else:
            ops.gemm16_lo8(hi, lo8, wt, sw, None, n, ops.EPI_F32, c=out, w8=w8)

    ref = [torch.empty(m, n, device="cuda") for _ in range(2)]
    for kind in range(2):
        run(kind, ref[kind])
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = {(i, kind): torch.empty(m, n, device="cuda") for i in range(2) for kind in range(2)}
    n_before = len(ops._workspaces)
    for rep in range(3):
        for i, st in enumerate((s1, s2)):
            with torch.cuda.stream(st):
                for kind in range(2):
                    run(kind, outs[(i, kind)])
    torch.cuda.synchronize()
    assert len(ops._workspaces) == n_before + 2                      # one workspace per new stream, created once
    ws = [ops._workspaces[(torch.cuda.current_device(), st.cuda_stream)] for st in (s1, s2)]
    assert ws[0] != ws[1]
    for (i, kind), o in outs.items():
        assert torch.equal(o, ref[kind]), f"stream {i}, kernel {kind}: result differs from the single-stream run"
    # without a workspace the plain entry point never runs a persistent variant, and says nothing else changed
    c = torch.empty(m, n, device="cuda")
    rc = _lib.lib().llark_gemm16_ex(30, 0, 1, 0, hi.data_ptr(), lo.data_ptr(), hi.stride(0), wt.data_ptr(), wt.stride(0), None, m, n,
                                    wt.shape[1], c.data_ptr(), c.stride(0), None, 0, None, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(c, ref[0])                                    # every tile variant computes the same product
