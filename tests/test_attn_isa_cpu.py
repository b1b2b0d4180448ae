"""ISA check of the training forward attention (csrc/llama.hip, attn_prefill_kernel<false, NQ, false, false>; no GPU).

Its softmax all-reduces the running maximum over the four lane groups of a query with v_permlane16_swap / v_permlane32_swap followed by a
max of the swapped pair.  Round 6 shipped that code for a few hours with `__builtin_bit_cast(float, a[1])` taken directly on the builtin's
vector result, which this hipcc reads as element 0: both casts became one register, the max folded away, and every lane used lane group
0's maximum -- a consistent but sub-maximal reference, an output ~1e-3 less accurate, and NO functional test failed until a 128 000-row
case tripped the per-row bar of tests/test_attn_bwd_gpu.py.  The property is visible in the generated code, so it is checked there: every
swap is followed by a v_max_f32 of its two registers."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_permlane_swaps_are_followed_by_the_max_of_the_pair(tmp_path):
    out = tmp_path / "llama.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", str(out),
                        os.path.join(ROOT, "llark_amd", "csrc", "llama.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = out.read_text().split("\n")
    found = 0
    for sym in ("_ZN5llark19attn_prefill_kernelILb0ELi2ELb0ELb0E", "_ZN5llark19attn_prefill_kernelILb0ELi1ELb0ELb0E"):
        start = next(i for i, l in enumerate(txt) if l.startswith(sym))
        end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith(".Lfunc_end"))
        body = [l.strip() for l in txt[start:end] if l.strip() and not l.strip().startswith((";", "."))]
        swaps = [i for i, l in enumerate(body) if l.startswith(("v_permlane16_swap", "v_permlane32_swap"))]
        assert len(swaps) >= 2 and len(swaps) % 2 == 0, (sym, len(swaps))
        for i in swaps:
            va, vb = [t.strip() for t in body[i].split(None, 1)[1].split(",")]
            window = body[i + 1: i + 6]
            ok = any(l.startswith("v_max_f32") and {va, vb} <= {t.strip() for t in l.split(None, 1)[1].split(",")[1:]} for l in window)
            assert ok, f"{sym}: `{body[i]}` is not followed by the max of {va} and {vb}: {window}"
            found += 1
    assert found >= 6           # two query sets x two swaps in the 128-query kernel, one set in the 64-query kernel
