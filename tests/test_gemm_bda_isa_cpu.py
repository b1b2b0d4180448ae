"""ISA audit of csrc/gemm_bda.hip (no GPU; hipcc cross-compiles here and on the GPU box).

The kernel's weight loads are inline asm that hipcc neither counts nor waits for (cdna_hip_programming.md 5.7 item 1): correctness
rests on properties of the GENERATED code that no functional test can establish -- a register of the in-flight ring that the
compiler copies, reuses or reads before the hand-placed s_waitcnt retires its load gives wrong results on some waves of some
launches only (both failure modes were hit while the kernel was written: the tail loads' dead destinations handed to the fragment
read-ahead; twelve v_mov of in-flight ring registers at the end of the prologue).  This test compiles the file to assembly and checks,
for every kernel in it:
  * no spills, no scratch;
  * the K loop and its peeled last iteration contain no v_mov, no scratch access and no s_waitcnt vmcnt(0) (no compiler drain);
  * from the issue of every global_load_dwordx4 until the s_waitcnt vmcnt(N) that retires it (in-order accounting over all VMEM
    operations, LDS-DMA included), no other instruction reads or writes its destination registers -- followed across the loop's
    back edge and into the peeled iteration.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _operands(line):
    body = line.split(None, 1)[1] if " " in line else ""
    return [t.strip() for t in re.split(r",\s*", body.split(" offset")[0].split(" offen")[0]) if t.strip()]


def _blocks(lines):
    blocks, order, cur = {"entry": []}, ["entry"], "entry"
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
        else:
            x = l.strip()
            if x and not x.startswith(";") and not x.startswith("."):
                blocks[cur].append(x)
    return blocks, order


def audit_kernels(txt, name_regex, min_kernels, shapes=((64, 8), (32, 4)), loads=8, tr_reads=32):
    """txt: lines of an AMDGPU assembly file; checks every kernel whose symbol matches name_regex (see the module docstring).
    shapes: (MFMAs, LDS-DMA requests) one K-step of the loop may hold; loads: its global_load_dwordx4 count; tr_reads: its transposing LDS
    reads when it has any.  Returns the number of (kernel, load) pairs followed to their retiring wait."""
    names = [l.split(":")[0] for l in txt if re.match(r"^" + name_regex + r"\S*:", l)]
    assert len(names) >= min_kernels, f"{len(names)} kernels match {name_regex}"
    total = 0
    for name in names:
        start = next(i for i, l in enumerate(txt) if l.startswith(name + ":"))
        end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith(".Lfunc_end"))
        blocks, order = _blocks(txt[start:end])
        hot = [b for b in order if (sum("v_mfma" in x for x in blocks[b]), sum(" lds" in x for x in blocks[b])) in shapes
               and sum("global_load_dwordx4" in x for x in blocks[b]) == loads]          # hi + lo form: 64 MFMAs + 8 DMA requests per K-step; plain: 32 + 4
        if not hot:
            continue                                                  # a kernel of the file that does not contain the DMA loop
        assert len(hot) in (1, 2), f"{name}: expected the K loop (+ a peeled last iteration), found {hot}"
        loop = next(b for b in hot if any(x.startswith("s_cbranch") and x.split()[-1] == b for x in blocks[b]))
        for b in hot:
            ins = blocks[b]
            assert not any(x.startswith("v_mov") or x.startswith("v_accvgpr") for x in ins), f"{name} {b}: register copies inside the K loop"
            assert not any(x.startswith("scratch_") for x in ins), f"{name} {b}: scratch access inside the K loop"
            assert not any(re.match(r"s_waitcnt.*vmcnt\(0\)", x) for x in ins), f"{name} {b}: a vmcnt(0) drain inside the K loop"
            if any(x.startswith("ds_read_b64_tr_b16") for x in ins):
                # the dW form (round 6): its transposing LDS reads are inline asm retired by hand-counted s_waitcnt lgkmcnt(N) -- exact only
                # while nothing else shares that counter inside the loop (scalar loads return out of order; other LDS operations shift the count)
                assert not any(x.startswith(("s_load", "s_buffer_load")) for x in ins), f"{name} {b}: a scalar load inside the K loop"
                assert all(x.startswith("ds_read_b64_tr_b16") for x in ins if x.startswith("ds_")), f"{name} {b}: an LDS operation the counts do not know"
                assert sum(x.startswith("ds_read_b64_tr_b16") for x in ins) == tr_reads, f"{name} {b}: expected {tr_reads} transposing reads"
        # the instruction stream a wave sees: loop body twice (back edge), then the peeled iteration (or the loop once more), then whatever follows
        tail_blocks = [b for b in hot if b != loop] or [loop]
        nxt = order.index(tail_blocks[-1]) + 1
        stream = blocks[loop] + blocks[loop] + sum((blocks[b] for b in tail_blocks), []) + (blocks[order[nxt]] if nxt < len(order) else [])
        vmem = [i for i, x in enumerate(stream) if x.startswith(("global_load", "buffer_load", "global_store", "buffer_store"))]
        checked = 0
        for i, x in enumerate(stream[: 2 * len(blocks[loop])]):
            if not x.startswith("global_load_dwordx4"):
                continue
            dest = _regs(_operands(x)[0])
            assert len(dest) == 4
            retired = None
            for j in range(i + 1, len(stream)):
                y = stream[j]
                m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", y)
                if m and sum(1 for v in vmem if i < v < j) <= int(m.group(1)):
                    retired = j
                    break
                touched = set()
                for tok in _operands(y):
                    touched |= _regs(tok)
                assert not (touched & dest), f"{name}: `{y}` touches v{sorted(touched & dest)} while `{x}` (stream position {i}) is in flight"
            assert retired is not None, f"{name}: load at stream position {i} is never retired"
            checked += 1
        assert checked == 2 * loads
        total += checked
    return total


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_gemm_bda_generated_code_keeps_the_inflight_ring_untouched(tmp_path):
    out = tmp_path / "bda.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-o", str(out), os.path.join(ROOT, "llark_amd", "csrc", "gemm_bda.hip")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", r.stderr)] + [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(spills) >= 20 and not any(spills), f"spills / scratch: {spills}"
    assert all(int(v) <= 256 for v in re.findall(r"VGPRs: (\d+)", r.stderr))
    txt = out.read_text().split("\n")
    assert audit_kernels(txt, "_ZN5llark15gemm_bda_kernel", 12) == 12 * 16       # 5 hi + lo and 7 plain epilogues (round 6: + the two training SwiGLU forms)
    assert audit_kernels(txt, "_ZN5llark18gemm_bda_ta_kernel", 2) == 2 * 16       # the dW form: contraction-major A through the transposing LDS read
    assert audit_kernels(txt, "_ZN5llark19gemm_bda_lnp_kernel", 2) == 2 * 16      # the LayerNorm-producer role on the same loop (fp16, bf16)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_gemm_bda16_generated_code_keeps_the_inflight_ring_untouched(tmp_path):
    """The dW product on the 16x16x32 MFMA shape (csrc/gemm_bda16.hip): the same hand-counted protocol with 128 MFMAs, 8 DMA requests,
    16 chunk loads and 64 transposing reads per K-step of 128 tokens."""
    out = tmp_path / "bda16.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-o", str(out), os.path.join(ROOT, "llark_amd", "csrc", "gemm_bda16.hip")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", r.stderr)] + [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(spills) >= 6 and not any(spills), f"spills / scratch: {spills}"
    assert all(int(v) <= 256 for v in re.findall(r"VGPRs: (\d+)", r.stderr))
    txt = out.read_text().split("\n")
    assert audit_kernels(txt, r"_ZN5llark12_GLOBAL__N_120gemm_bda16_ta_kernel", 2, shapes=((128, 8),), loads=16, tr_reads=64) == 2 * 32


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_gemm_bd_sk_generated_code_keeps_the_inflight_ring_untouched(tmp_path):
    """The same audit on the K-cut / stream-K kernels (csrc/gemm_bd_sk.hip), which inline the same hand-counted loop and are the default for
    the Llama o_proj / down_proj products (until round 6 they sat inside gemm.hip, a five-minute compile, and were audited by
    hand only).  Twelve bf16 instantiations (hi + lo and plain x six epilogues) carry the DMA loop; the fp16 ones keep the register-staged
    loop and are skipped by the audit."""
    out = tmp_path / "sk.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-o", str(out), os.path.join(ROOT, "llark_amd", "csrc", "gemm_bd_sk.hip")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", r.stderr)] + [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(spills) >= 24 and not any(spills), f"spills / scratch: {spills}"
    txt = out.read_text().split("\n")
    assert audit_kernels(txt, "_ZN5llark17gemm_bd_sk_kernel", 24) == 12 * 16


if __name__ == "__main__":      # python tests/test_gemm_bda_isa_cpu.py <file.s> <kernel symbol regex>: the same audit on any assembly file
    import sys

    print("loads followed to their retiring wait:", audit_kernels(open(sys.argv[1]).read().split("\n"), sys.argv[2], 1))
