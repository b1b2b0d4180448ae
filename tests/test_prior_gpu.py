"""GPU parity: prior kernels (embed, LayerNorm-split, split-fp16 GEMM, factored attention, pooling)
and the whole prior forward vs the fp32 CPU oracle.

Tolerances (floating point): the reference runs the prior in fp32 with fp16-valued weights
(fp16=False); the HIP path carries activations as hi+lo fp16 (22 bits) into fp32-accumulating MFMA.
Bound used for end-to-end activations / embeddings: max|err| <= 1e-4 * max|ref| (BASELINE.json's
1e-4), individual kernels much tighter (stated per test)."""
import math

import numpy as np
import pytest
import torch

from conftest import report_close
from llark_amd.jukebox.hparams import hparams_5b_depth, hparams_tiny
from llark_amd.jukebox.synthetic import make_prior_weights

pytestmark = pytest.mark.gpu


def test_split16_and_pack():
    from llark_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 70, generator=g) * 3
    hi, lo = ops.split16(x.cuda(), torch.float16)
    assert hi.shape == (37, 96)
    rh = x.half()
    rl = (x - rh.float()).half()
    assert torch.equal(hi[:, :70].cpu(), rh) and torch.equal(lo[:, :70].cpu(), rl)
    assert (hi[:, 70:] == 0).all() and (lo[:, 70:] == 0).all()
    w = (torch.randn(70, 50, generator=g)).half()
    wt = ops.pack_weight16(w.cuda(), True, torch.float16)
    assert wt.shape == (50, 96) and torch.equal(wt[:, :70].cpu(), w.t()) and (wt[:, 70:] == 0).all()
    wl = torch.randn(50, 70, generator=g)
    wt2 = ops.pack_weight16(wl.cuda(), False, torch.bfloat16)
    assert torch.equal(wt2[:, :70].cpu(), wl.bfloat16())


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (64, 96, 64), (200, 150, 70), (1024, 3600, 1200), (515, 4800, 4800)])
def test_gemm_split_f16(m, n, k):
    """C = A.W + b in split mode: fp32-class accuracy. Bound: 4e-7*(|A|.|W|) (K-ordered fp32 accumulate)."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.05).half()
    b = torch.randn(n, generator=g)
    hi, lo = ops.split16(a.cuda(), torch.float16)
    wt = ops.pack_weight16(w.cuda(), True, torch.float16)
    c = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16(hi, lo, wt, b.cuda(), n, ops.EPI_F32, c=c)
    ref = a.double() @ w.double() + b.double()
    bound = (a.abs().double() @ w.abs().double())
    err = (c.cpu().double() - ref).abs()
    assert torch.isfinite(c).all(), "non-finite or unwritten outputs"
    worst = (err / (bound + 1e-30)).max().item()
    assert worst < 6e-7, f"split gemm {m}x{n}x{k}: err/(|A||W|) = {worst:.3e}; max err {err.max():.3e}"
    # single pass (hi only) must be fp16-class, and clearly worse than split (proves lo is used)
    c1 = torch.empty((m, n), device="cuda")
    ops.gemm16(hi, None, wt, b.cuda(), n, ops.EPI_F32, c=c1)
    err1 = (c1.cpu().double() - ref).abs()
    assert (err1 / (bound + 1e-30)).max().item() < 1e-3
    if k >= 64:
        assert err1.max() > 8 * err.max()


@pytest.mark.parametrize("variant", [0, 1, 2, 11, 12, 20])
def test_gemm_tile_variants(variant):
    """Every tile variant computes the same product (ragged M/N, K padded to 64)."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(variant)
    m, n, k = 333, 450, 200
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.1).half()
    b = torch.randn(n, generator=g)
    hi, lo = ops.split16(a.cuda(), torch.float16, kmult=64)
    wt = ops.pack_weight16(w.cuda(), True, torch.float16, kmult=64)
    c = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16(hi, lo, wt, b.cuda(), n, ops.EPI_F32, c=c, variant=variant)
    ref = a.double() @ w.double() + b.double()
    worst = ((c.cpu().double() - ref).abs() / (a.abs().double() @ w.abs().double())).max().item()
    assert worst < 6e-7, worst


@pytest.mark.parametrize("v256", [31, 30], ids=["v31-phases-over-N", "v30-M-split-ring"])
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (1000, 3600, 1216), (2048, 4800, 4800), (515, 290, 200), (8192, 1200, 640),
                                   (40000, 600, 448), (70000, 2500, 192)])
def test_gemm256_ring_tile_bit_identical(m, n, k, v256):
    """The 256x256x64 split tiles -- csrc/gemm256n.hip (variant 31, the default of the prior: phases over N, resident A fragments,
    A ring of 7 quarter units, wave pairs half a phase apart) and csrc/gemm256.hip (variant 30: round 2's M-split LDS ring) --
    accumulate every output element in the same order as the 128x256 LDS-staged kernel (k16 steps ascending, hi pass then lo
    pass): they must agree BIT FOR BIT on ragged M / N (row clamping, masked stores), 2 .. 75 K-steps (ring wrap, tail waits;
    3 and 7 K-steps walk the 7-slot ring through every residue), more tiles than CUs (tile-to-tile overlap, chunk barriers),
    every epilogue the prior uses, fp16 and bf16 -- and against an fp64 reference within the split scheme's bound."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(m * 7 + n + k)
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.1).half()
    b = torch.randn(n, generator=g).cuda()
    r = torch.randn(m, n, generator=g).cuda()
    for dt in (torch.float16, torch.bfloat16):
        hi, lo = ops.split16(a.cuda(), dt, kmult=64)
        wt = ops.pack_weight16(w.cuda(), True, dt, kmult=64)
        c0 = torch.full((m, n), float("nan"), device="cuda")
        c1 = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_F32, c=c0, variant=12)
        for rep in range(3):                                             # repeated launches: a race in the ring would come and go
            c1.fill_(float("nan"))
            ops.gemm16(hi, lo, wt, b, n, ops.EPI_F32, c=c1, variant=v256)
            assert torch.equal(c0, c1), f"F32 {dt} rep {rep}: {int((c0 != c1).sum())} elements differ, max diff {(c0 - c1).abs().nan_to_num(1e9).max().item():.3e}"
        a16 = hi.float().cpu()[:, :k].double() + lo.float().cpu()[:, :k].double()
        w16 = wt.float().cpu()[:n, :k].double()
        ref = a16 @ w16.t() + b.cpu().double()
        bound = 2e-6 * (a16.abs() @ w16.abs().t()) + 1e-6
        assert bool(((c1.cpu().double() - ref).abs() <= bound).all()), "gemm256 vs fp64 reference out of the split-scheme bound"
        c0, c1 = r.clone(), r.clone()
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_RESID, c=c0, resid=c0, variant=12)
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_RESID, c=c1, resid=c1, variant=v256)
        assert torch.equal(c0, c1), "RESID epilogue differs"
        o0 = [torch.zeros((m, n), dtype=dt, device="cuda") for _ in range(2)]
        o1 = [torch.zeros((m, n), dtype=dt, device="cuda") for _ in range(2)]
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_QGELU_SPLIT, out_hi=o0[0], out_lo=o0[1], variant=12)
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_QGELU_SPLIT, out_hi=o1[0], out_lo=o1[1], variant=v256)
        assert torch.equal(o0[0], o1[0]) and torch.equal(o0[1], o1[1]), "QGELU_SPLIT epilogue differs"
    # non-split operands are not this kernel's: variant 30 must fall back (and still be right), not fail
    c2 = torch.full((m, n), float("nan"), device="cuda")
    c3 = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16(hi, None, wt, b, n, ops.EPI_F32, c=c2, variant=v256)
    ops.gemm16(hi, None, wt, b, n, ops.EPI_F32, c=c3, variant=12)
    assert torch.equal(c2, c3)


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (1000, 3600, 1216), (2048, 4800, 4800), (515, 292, 200), (8192, 1200, 640),
                                   (40000, 600, 448), (70000, 2500, 192), (300, 290, 256)])
def test_gemm256x_16x16x32_tile(m, n, k):
    """csrc/gemm256x.hip (variant 32, round 4: the prior's default): gemm256n's tile, rings and LDS-DMA protocol on
    v_mfma_f32_16x16x32, product computed transposed, 16-byte epilogue stores.  The instruction sums 32 products per step, so it
    is NOT bit-identical to the 32x32x16 kernels: checked against an fp64 reference within the split scheme's bound, against the
    128x256 kernel to rounding, run-to-run bit-equal (a race in the rings would come and go), on ragged M / N, 2 .. 75 K-steps,
    more tiles than CUs, every epilogue the prior uses, fp16 and bf16; n = 290 (n % 4 != 0) must fall back and stay right."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(m * 7 + n + k)
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.1).half()
    b = torch.randn(n, generator=g).cuda()
    r = torch.randn(m, n, generator=g).cuda()
    for dt in (torch.float16, torch.bfloat16):
        hi, lo = ops.split16(a.cuda(), dt, kmult=64)
        wt = ops.pack_weight16(w.cuda(), True, dt, kmult=64)
        a16 = hi.float().cpu()[:, :k].double() + lo.float().cpu()[:, :k].double()
        w16 = wt.float().cpu()[:n, :k].double()
        ref = a16 @ w16.t() + b.cpu().double()
        bound = 2e-6 * (a16.abs() @ w16.abs().t()) + 1e-6
        c0 = torch.full((m, n), float("nan"), device="cuda")
        c1 = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_F32, c=c0, variant=12)
        first = None
        for rep in range(3):
            c1.fill_(float("nan"))
            ops.gemm16(hi, lo, wt, b, n, ops.EPI_F32, c=c1, variant=32)
            if first is None:
                first = c1.clone()
            assert torch.equal(first, c1), f"F32 {dt}: launch {rep} differs from launch 0 in {int((first != c1).sum())} elements"
        err = (c1.cpu().double() - ref).abs()
        assert bool((err <= bound).all()), f"gemm256x {dt} vs fp64 reference: worst excess {float((err - bound).max()):.3e}"
        assert float((c1 - c0).abs().max()) <= 4e-6 * float(ref.abs().max()) + 1e-6
        c1 = r.clone()
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_RESID, c=c1, resid=c1, variant=32)                      # in place, like h += ...
        assert bool(((c1.cpu().double() - (ref + r.cpu().double())).abs() <= bound + 1e-6).all()), "RESID epilogue"
        o1 = [torch.full((m, n + 4), float("nan"), dtype=dt, device="cuda") for _ in range(2)]
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_QGELU_SPLIT, out_hi=o1[0], out_lo=o1[1], variant=32)
        got = o1[0][:, :n].float().cpu().double() + o1[1][:, :n].float().cpu().double()
        want = ref * torch.sigmoid(1.702 * ref)
        q = (2.0 ** -21 if dt == torch.float16 else 2.0 ** -16) * want.abs()                         # what a hi + lo pair of planes cannot carry
        assert bool(((got - want).abs() <= 3.0 * bound + q + 2e-6 * want.abs() + 1e-6).all()), "QGELU_SPLIT epilogue"
        assert bool(torch.isnan(o1[0][:, n:]).all() and torch.isnan(o1[1][:, n:]).all()), "QGELU_SPLIT wrote past column n"
        o0 = [torch.zeros((m, n + 4), dtype=dt, device="cuda") for _ in range(2)]
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_QGELU_SPLIT, out_hi=o0[0], out_lo=o0[1], variant=12)
        d = (o0[0][:, :n].float() + o0[1][:, :n].float()) - (o1[0][:, :n].float() + o1[1][:, :n].float())
        assert float(d.abs().max()) <= 2.0 * float((3.0 * bound + q).max()) + 1e-6               # two correct kernels, each inside the bound


@pytest.mark.parametrize("m,n,k", [(8192, 4800, 192), (8200, 3648, 256), (20000, 1216, 320)])
def test_gemm16_ln_folded_layernorm_roles(m, n, k):
    """llark_gemm16_ln (round 4): the LayerNorm folded into the 256x256 tile's epilogues.
    Producer (EPI_RESID): the stream it writes is BIT-equal to the plain product's; the planes are the hi / lo split of c . gamma;
    the per-slice sums, reduced by llark_ln_stats_finalize, are the row's mean and 1 / sqrt(var + eps); run-to-run bit-equal.
    Consumer: rstd (acc - mean gw) + bw  ==  LayerNorm(c) W + b evaluated in float64, inside the split scheme's bound, for the
    F32 and the QGELU_SPLIT epilogue; ragged M, ragged last column tile; shapes the tile does not take are refused."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(m + 3 * n + k)
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.2).half()
    b = torch.randn(n, generator=g).cuda()
    r = (torch.randn(m, n, generator=g) * 3.0 + 0.7).cuda()             # a row mean that is not ~0: the -mean.gw term matters
    gamma = (1.0 + 0.3 * torch.randn(n, generator=g)).cuda()
    beta = (0.2 * torch.randn(n, generator=g)).cuda()
    assert ops.gemm16_ln_takes(m, n, ops.round_up(k, 64)) and not ops.gemm16_ln_takes(512, 512, 256)
    hi, lo = ops.split16(a.cuda(), torch.float16, kmult=64)
    wt = ops.pack_weight16(w.cuda(), True, torch.float16, kmult=64)
    # ---- producer
    c0 = r.clone()
    ops.gemm16(hi, lo, wt, b, n, ops.EPI_RESID, c=c0, resid=c0, variant=32)
    nparts = 2 * ((n + 255) // 256)
    first = None
    for rep in range(2):
        c1 = r.clone()
        ph, pl = (torch.full((m, n + 8), float("nan"), dtype=torch.float16, device="cuda") for _ in range(2))
        part = torch.full((m, nparts, 2), float("nan"), device="cuda")
        stat = torch.full((m, 2), float("nan"), device="cuda")
        ops.gemm16_ln(hi, lo, wt, b, n, ops.EPI_RESID, gamma, ln_part=part, c=c1, resid=c1, out_hi=ph, out_lo=pl)
        ops.ln_stats_finalize(part, m, nparts, n, 1e-5, stat)
        if first is None:
            first = (c1.clone(), ph.clone(), pl.clone(), stat.clone())
    assert torch.equal(c1, c0), f"producer stream differs from the plain RESID product in {int((c1 != c0).sum())} elements"
    assert torch.equal(first[0], c1) and torch.equal(first[1][:, :n], ph[:, :n]) and torch.equal(first[2][:, :n], pl[:, :n]) and torch.equal(first[3], stat)
    assert bool(torch.isnan(ph[:, n:]).all() and torch.isnan(pl[:, n:]).all()), "producer wrote planes past column n"
    xg = c1 * gamma
    want_hi = xg.half()
    # (bit-equal: found the toolchain's v_fma_mixlo_f16 double-rounding mismatch in round 4 -- csrc/gemm256x.hip fp_pin)
    assert torch.equal(ph[:, :n], want_hi) and torch.equal(pl[:, :n], (xg - want_hi.float()).half()), "planes are not split16(c . gamma)"
    c64 = c1.double()
    mean64, var64 = c64.mean(1), c64.var(1, unbiased=False)
    assert float((stat[:, 0].double() - mean64).abs().max()) <= 2e-6 * float(c64.abs().max())
    rstd64 = 1.0 / torch.sqrt(var64 + 1e-5)
    assert float((stat[:, 1].double() / rstd64 - 1.0).abs().max()) <= 3e-6
    # ---- consumer: a second product whose operand is LayerNorm(c1); K = n, N = n2
    n2 = 4800 if m < 16384 else 2048
    w2 = (torch.randn(n, n2, generator=g) * 0.05).half()
    b2 = torch.randn(n2, generator=g).cuda()
    wt2 = ops.pack_weight16(w2.cuda(), True, torch.float16, kmult=64)
    kp2 = wt2.shape[1]
    if ph.shape[1] < kp2 or not ops.gemm16_ln_takes(m, n2, kp2):
        return
    w64 = wt2[:, :n].double()
    gw = (w64 @ gamma.double()).float()
    bw = (w64 @ beta.double() + b2.double()).float()
    ph0, pl0 = (torch.zeros((m, kp2), dtype=torch.float16, device="cuda") for _ in range(2))
    ph0[:, :n], pl0[:, :n] = ph[:, :n], pl[:, :n]
    ln64 = (c64 - mean64[:, None]) * rstd64[:, None] * gamma.double() + beta.double()
    ref = ln64 @ w64.t() + b2.double()
    # the error of the form: the planes carry x . gamma to 2^-21, the products sum in fp32, the mean term cancels against the sum
    xg_abs = (c64 * gamma.double()).abs()
    bound = rstd64[:, None] * (3e-6 * (xg_abs @ w64.abs().t()) + 3e-6 * mean64.abs()[:, None] * gw.double().abs()) + 2e-6 * ref.abs() + 1e-6
    out = torch.full((m, n2), float("nan"), device="cuda")
    ops.gemm16_ln(ph0, pl0, wt2, bw, n2, ops.EPI_F32, gw, ln_stat=stat, c=out)
    err = (out.double() - ref).abs()
    assert bool((err <= bound).all()), f"consumer F32 vs float64 LayerNorm + product: worst excess {float((err - bound).max()):.3e}, worst err {float(err.max()):.3e}"
    # and against the unfolded path (LayerNorm kernel + plain product): two correct evaluations of the same thing
    lh, ll = (torch.zeros((m, kp2), dtype=torch.float16, device="cuda") for _ in range(2))
    ops.layernorm_split(c1, gamma, beta, 1e-5, lh, ll)
    out_u = torch.empty((m, n2), device="cuda")
    ops.gemm16(lh, ll, wt2, b2, n2, ops.EPI_F32, c=out_u, variant=32)
    print(f"\n[ln-fold] m={m} k={n} n={n2}: folded vs float64 {float(err.max()):.2e}, unfolded vs float64 {float((out_u.double() - ref).abs().max()):.2e}, "
          f"max|out| {float(ref.abs().max()):.1f}")
    oh, ol = (torch.full((m, n2 + 8), float("nan"), dtype=torch.float16, device="cuda") for _ in range(2))
    ops.gemm16_ln(ph0, pl0, wt2, bw, n2, ops.EPI_QGELU_SPLIT, gw, ln_stat=stat, out_hi=oh, out_lo=ol)
    got = oh[:, :n2].double() + ol[:, :n2].double()
    want = ref * torch.sigmoid(1.702 * ref)
    assert bool(((got - want).abs() <= 1.2 * bound + 2.0 ** -21 * want.abs() + 2e-6 * want.abs() + 1e-6).all()), "consumer QGELU_SPLIT"
    assert bool(torch.isnan(oh[:, n2:]).all()), "consumer wrote past column n"
    # refused, not mis-computed
    with pytest.raises(RuntimeError):
        ops.gemm16_ln(ph0[:512], pl0[:512], wt2, bw, n2, ops.EPI_F32, gw, ln_stat=stat, c=out[:512])


@pytest.mark.parametrize("m,n,k,dt", [(8232, 4800, 1216, torch.float16), (640, 1216, 256, torch.float16), (300, 4800, 192, torch.bfloat16)])
def test_gemm16_lnp_fragw_producer_on_the_dma_loop(m, n, k, dt):
    """llark_gemm16_lnp_fragw (round 5): the LayerNorm PRODUCER role on gemm_bda's 128x256 tiles (two workgroups to a CU, product
    computed transposed, fragment-major weights) -- the prior's attention-output product.  Against float64 inside the split scheme's
    bound and against the persistent tile's producer to rounding (the two MFMA shapes sum in different orders); the planes are
    BIT-equal to split16(((c - shift) scale) gamma) of the stream the kernel itself wrote; the 64-column partial sums reduce to the
    row's statistics; in place (resid aliases c); ragged last row tile, ragged last column tile (a whole wave beyond n), pad columns
    untouched; run-to-run bit-equal; refused below three K-steps."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.1).half()
    b = torch.randn(n, generator=g).cuda()
    r = (torch.randn(m, n, generator=g) * 2.0 + 0.5).cuda()
    gamma = (1.0 + 0.3 * torch.randn(n, generator=g)).cuda()
    hi, lo = ops.split16(a.cuda(), dt, kmult=64)
    wt = ops.pack_weight16(w.cuda(), True, dt, kmult=64)
    kp = wt.shape[1]
    wf = ops.pack_weight16_frag(wt, n)
    a16 = hi.float().cpu()[:, :k].double() + lo.float().cpu()[:, :k].double()
    w16 = wt.float().cpu()[:n, :k].double()
    ref = a16 @ w16.t() + b.cpu().double() + r.cpu().double()
    bound = 2e-6 * (a16.abs() @ w16.abs().t()) + 2e-7 * ref.abs() + 1e-6
    pred0 = torch.empty((m, 2), device="cuda")
    ops.ln_row_pred(r, 1e-5, pred0)
    for use_pred in (True, False):
        first = None
        for rep in range(2):
            c1 = r.clone()
            pred = pred0.clone() if use_pred else None
            ph, pl = (torch.full((m, n + 8), float("nan"), dtype=dt, device="cuda") for _ in range(2))
            part = torch.full((m, (n + 63) // 64, 2), float("nan"), device="cuda")
            stat = torch.full((m, 2), float("nan"), device="cuda")
            nparts = ops.gemm16_lnp_fragw(hi, lo, wf, b, n, kp, gamma, part, c1, c1, ph, pl, ln_pred=pred)
            assert nparts == (n + 63) // 64
            ops.ln_stats_finalize(part, m, nparts, n, 1e-5, stat, pred)
            if first is None:
                first = (c1.clone(), ph[:, :n].clone(), pl[:, :n].clone(), stat.clone())
        assert all(torch.equal(x, y) for x, y in zip(first, (c1, ph[:, :n], pl[:, :n], stat))), "not run-to-run bit-equal"
        err = (c1.cpu().double() - ref).abs()
        assert bool((err <= bound).all()), f"stream vs float64: worst excess {float((err - bound).max()):.3e}"
        assert bool(torch.isnan(ph[:, n:]).all() and torch.isnan(pl[:, n:]).all()), "planes written past column n"
        assert bool(torch.isfinite(part).all()), "a partial sum was not written"
        shift = pred0[:, 0:1] if use_pred else torch.zeros((m, 1), device="cuda")
        scale = pred0[:, 1:2] if use_pred else torch.ones((m, 1), device="cuda")
        xg = ((c1 - shift) * scale) * gamma
        want_hi = xg.to(dt)
        assert torch.equal(ph[:, :n], want_hi) and torch.equal(pl[:, :n], (xg - want_hi.float()).to(dt)), "planes are not split16(((c - shift) scale) gamma)"
        c64 = c1.double()
        mean64, var64 = c64.mean(1), c64.var(1, unbiased=False)
        sd, rstd64 = torch.sqrt(var64 + 1e-5), 1.0 / torch.sqrt(var64 + 1e-5)
        assert float(((stat[:, 0].double() / scale[:, 0].double() + shift[:, 0].double() - mean64).abs() / sd).max()) <= 1e-5
        assert float((stat[:, 1].double() * scale[:, 0].double() / rstd64 - 1.0).abs().max()) <= 1e-5
        if use_pred:
            assert float(((pred[:, 0].double() - mean64).abs() / sd).max()) <= 1e-5, "pred was not replaced by the measured mean"
    if dt == torch.float16 and ops.gemm16_ln_takes(m, n, kp):        # the persistent tile's producer: same values to rounding
        c2 = r.clone()
        ph2, pl2 = (torch.zeros((m, n + 8), dtype=dt, device="cuda") for _ in range(2))
        part2 = torch.empty((m, 2 * ((n + 255) // 256), 2), device="cuda")
        ops.gemm16_ln(hi, lo, wt, b, n, ops.EPI_RESID, gamma, ln_part=part2, c=c2, resid=c2, out_hi=ph2, out_lo=pl2)
        assert float((c2 - c1).abs().max()) <= 4e-6 * float(ref.abs().max()) + 1e-6
    with pytest.raises(RuntimeError):                                 # two K-steps: below the ring's prologue
        ops.gemm16_lnp_fragw(hi[:, :128].contiguous(), lo[:, :128].contiguous(), wf[: ops.round_up(n, 32) * 128], b, n, 128, gamma, part, c1, c1, ph, pl)


def test_gemm16_ln_predicted_statistics_adversarial_rows():
    """llark_gemm16_ln_p / llark_ln_stats_finalize_p / llark_ln_row_pred (round 5): the folded LayerNorm with planes
    pre-normalised by PREDICTED row statistics, on rows the unscaled planes lose bits on -- std 1e-3 (fp16 lo plane subnormal), std 1e3
    (hi plane near overflow), mean = 50 sigma, and ordinary rows -- against float64 and against the unfolded path (LayerNorm kernel +
    plain product).  Checks: the stream c is BIT-equal to the plain product's; the planes are split16(((c - shift) scale) gamma);
    stat = ((mean - shift) scale, rstd / scale); pred is replaced by (mean, nearest power of two of rstd); the consumer (unchanged
    kernel) is as close to float64 as the unfolded path on EVERY row class, where round 4's unscaled planes are 8x further out on
    the small rows; run-to-run bit-equal."""
    from llark_amd import ops
    m, n, k = 8192, 4800, 192
    g = torch.Generator().manual_seed(99)
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.2).half()
    b = torch.randn(n, generator=g).cuda()
    r = torch.randn(m, n, generator=g) * 3.0
    cls = torch.arange(m) % 4                          # 0 ordinary, 1 std ~1e-3, 2 std ~1e3, 3 mean 50 sigma
    a[cls == 1] *= 3e-4
    r[cls == 1] *= 3e-4
    a[cls == 2] *= 300.0
    r[cls == 2] *= 300.0
    r[cls == 3] += 50.0 * 4.0
    r = r.cuda()
    bsel = torch.zeros_like(b)                          # an N(0, 1) bias would lift the 1e-3 rows back to order one
    gamma = (1.0 + 0.3 * torch.randn(n, generator=g)).cuda()
    beta = (0.2 * torch.randn(n, generator=g)).cuda()
    hi, lo = ops.split16(a.cuda(), torch.float16, kmult=64)
    wt = ops.pack_weight16(w.cuda(), True, torch.float16, kmult=64)
    c0 = r.clone()
    ops.gemm16(hi, lo, wt, bsel, n, ops.EPI_RESID, c=c0, resid=c0, variant=32)
    # prediction from a DIFFERENT (earlier) state of the rows: the residual before this product -- off, as in the real chain
    pred0 = torch.empty((m, 2), device="cuda")
    ops.ln_row_pred(r, 1e-5, pred0)
    r64 = r.double()
    assert float((pred0[:, 0].double() - r64.mean(1)).abs().max()) <= 2e-6 * float(r64.abs().max())
    ratio = pred0[:, 1].double() * torch.sqrt(r64.var(1, unbiased=False) + 1e-5)
    assert float(ratio.max()) <= 1.4143 and float(ratio.min()) >= 0.7071, "scale is not the nearest power of two of rstd"
    assert bool((torch.frexp(pred0[:, 1])[0] == 0.5).all()), "scale must be an exact power of two"
    nparts = 2 * ((n + 255) // 256)
    first = None
    for rep in range(2):
        c1 = r.clone()
        pred = pred0.clone()
        ph, pl = (torch.full((m, n + 8), float("nan"), dtype=torch.float16, device="cuda") for _ in range(2))
        part = torch.full((m, nparts, 2), float("nan"), device="cuda")
        stat = torch.full((m, 2), float("nan"), device="cuda")
        ops.gemm16_ln(hi, lo, wt, bsel, n, ops.EPI_RESID, gamma, ln_part=part, c=c1, resid=c1, out_hi=ph, out_lo=pl, ln_pred=pred)
        ops.ln_stats_finalize(part, m, nparts, n, 1e-5, stat, pred)
        if first is None:
            first = (c1.clone(), ph[:, :n].clone(), pl[:, :n].clone(), stat.clone(), pred.clone())
    assert torch.equal(c1, c0), "producer stream differs from the plain RESID product"
    assert all(torch.equal(x, y) for x, y in zip(first, (c1, ph[:, :n], pl[:, :n], stat, pred))), "not run-to-run bit-equal"
    assert bool(torch.isnan(ph[:, n:]).all() and torch.isnan(pl[:, n:]).all()), "producer wrote planes past column n"
    shift, scale = pred0[:, 0:1], pred0[:, 1:2]
    xg = ((c1 - shift) * scale) * gamma
    want_hi = xg.half()
    assert torch.isfinite(want_hi.float()).all()
    assert torch.equal(ph[:, :n], want_hi) and torch.equal(pl[:, :n], (xg - want_hi.float()).half()), "planes are not split16(((c - shift) scale) gamma)"
    c64 = c1.double()
    mean64, var64 = c64.mean(1), c64.var(1, unbiased=False)
    rstd64 = 1.0 / torch.sqrt(var64 + 1e-5)
    sd = torch.sqrt(var64 + 1e-5)
    assert float(((stat[:, 0].double() / scale[:, 0].double() + shift[:, 0].double() - mean64).abs() / sd).max()) <= 1e-5, "stat.x != (mean - shift) scale"
    assert float((stat[:, 1].double() * scale[:, 0].double() / rstd64 - 1.0).abs().max()) <= 1e-5, "stat.y != rstd / scale"
    assert float(((pred[:, 0].double() - mean64).abs() / sd).max()) <= 1e-5, "pred was not replaced by the measured mean"
    rr = pred[:, 1].double() / rstd64
    assert float(rr.max()) <= 1.4143 and float(rr.min()) >= 0.7071
    # ---- consumer (the round-4 kernel, unchanged) on the pre-normalised planes vs round 4's planes vs the unfolded path
    n2 = 4800
    w2 = (torch.randn(n, n2, generator=g) * 0.05).half()
    b2 = torch.randn(n2, generator=g).cuda()
    wt2 = ops.pack_weight16(w2.cuda(), True, torch.float16, kmult=64)
    kp2 = wt2.shape[1]
    assert ops.gemm16_ln_takes(m, n2, kp2)
    w64 = wt2[:, :n].double()
    gw = (w64 @ gamma.double()).float()
    bw = (w64 @ beta.double() + b2.double()).float()
    ref = ((c64 - mean64[:, None]) * rstd64[:, None] * gamma.double() + beta.double()) @ w64.t() + b2.double()

    def consume(planes_hi, planes_lo, st):
        p0, p1 = (torch.zeros((m, kp2), dtype=torch.float16, device="cuda") for _ in range(2))
        p0[:, :n], p1[:, :n] = planes_hi[:, :n], planes_lo[:, :n]
        out = torch.full((m, n2), float("nan"), device="cuda")
        ops.gemm16_ln(p0, p1, wt2, bw, n2, ops.EPI_F32, gw, ln_stat=st, c=out)
        return out.double()

    out_p = consume(ph, pl, stat)
    # round 4's form on the same rows (no prediction)
    c2 = r.clone()
    ph4, pl4 = (torch.zeros((m, n + 8), dtype=torch.float16, device="cuda") for _ in range(2))
    part4, stat4 = torch.zeros((m, nparts, 2), device="cuda"), torch.zeros((m, 2), device="cuda")
    ops.gemm16_ln(hi, lo, wt, bsel, n, ops.EPI_RESID, gamma, ln_part=part4, c=c2, resid=c2, out_hi=ph4, out_lo=pl4)
    ops.ln_stats_finalize(part4, m, nparts, n, 1e-5, stat4)
    out_4 = consume(ph4, pl4, stat4)
    lh, ll = (torch.zeros((m, kp2), dtype=torch.float16, device="cuda") for _ in range(2))
    ops.layernorm_split(c1, gamma, beta, 1e-5, lh, ll)
    out_u = torch.empty((m, n2), device="cuda")
    ops.gemm16(lh, ll, wt2, b2, n2, ops.EPI_F32, c=out_u, variant=32)
    scale_out = float(ref.abs().max())
    names = ("ordinary", "std 1e-3", "std 1e3", "mean 50 sigma")
    for ci, nm in enumerate(names):
        sel = (cls == ci).cuda()
        e_p = float((out_p[sel] - ref[sel]).abs().max()) / scale_out
        e_4 = float((out_4[sel] - ref[sel]).abs().nan_to_num(float("inf")).max()) / scale_out
        e_u = float((out_u[sel].double() - ref[sel]).abs().max()) / scale_out
        print(f"\n[ln-pred] rows {nm:14s}: predicted-statistics planes {e_p:.2e} | round-4 planes {e_4:.2e} | unfolded {e_u:.2e}  (of max|out| {scale_out:.1f})")
        assert e_p <= max(3.0 * e_u, 1e-6), f"{nm}: predicted-statistics fold {e_p:.2e} vs unfolded {e_u:.2e}"
    sel = (cls == 1).cuda()
    assert float((out_4[sel] - ref[sel]).abs().max()) >= 4.0 * float((out_p[sel] - ref[sel]).abs().max()), \
        "expected round 4's unscaled planes to be several times further from float64 on the std 1e-3 rows (measured 8x)"


@pytest.mark.parametrize("m,n,k", [(2968, 768, 4096), (371, 512, 192), (1000, 832, 1216), (130, 64, 256)])
def test_gemm_bda_lds_dma_a_operand_bit_identical(m, n, k):
    """csrc/gemm_bda.hip (round 5; llark_gemm16_fragw variant 2): the hi + lo B-direct product with A staged by LDS-DMA, a second
    set of A fragments read one sub-step ahead and hand-counted weight loads -- BIT-identical to variant 0 (gemm_bd_kernel) for
    every epilogue it takes: ragged M (row clamping in the DMA source addresses, masked stores: 2968 = 23 x 128 + 24), ragged N,
    the shortest K it accepts (3 K-steps = the ring's prologue), and repeated launches (the hand-counted waits must not depend
    on what is in flight from the previous launch).  Shapes outside its domain are refused, not mis-computed."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(m + 2 * n + k)
    a = torch.randn(m, k, generator=g)
    wb = (torch.randn(n, k, generator=g) * 0.1).bfloat16()
    b = torch.randn(n, generator=g).cuda()
    r = torch.randn(m, n, generator=g).cuda()
    hi, lo = ops.split16(a.cuda(), torch.bfloat16, kmult=64)
    wt = ops.pack_weight16(wb.cuda(), False, torch.bfloat16, kmult=64)
    wf = ops.pack_weight16_frag(wt, n)
    kp = wt.shape[1]
    for rep in range(3):
        c0 = torch.full((m, n), float("nan"), device="cuda")
        c1 = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16_fragw(hi, lo, wf, b, n, kp, ops.EPI_F32, c=c0, variant=0, stream_k=False)
        ops.gemm16_fragw(hi, lo, wf, b, n, kp, ops.EPI_F32, c=c1, variant=2)
        assert torch.equal(c0, c1), f"F32 rep {rep}: {int((c0 != c1).sum())} elements differ, max {(c0 - c1).abs().nan_to_num(9e9).max().item():.3e}"
    ref = (hi.double()[:, :k] + lo.double()[:, :k]) @ wt.double()[:n, :k].t() + b.double()
    assert float((c1.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4
    c0, c1 = r.clone(), r.clone()
    ops.gemm16_fragw(hi, lo, wf, b, n, kp, ops.EPI_RESID, c=c0, resid=c0, variant=0, stream_k=False)
    ops.gemm16_fragw(hi, lo, wf, b, n, kp, ops.EPI_RESID, c=c1, resid=c1, variant=2)
    assert torch.equal(c0, c1)
    o0 = [torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    o1 = [torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    ops.gemm16_fragw(hi, lo, wf, b, n, kp, ops.EPI_SPLIT16, out_hi=o0[0], out_lo=o0[1], variant=0, stream_k=False)
    ops.gemm16_fragw(hi, lo, wf, b, n, kp, ops.EPI_SPLIT16, out_hi=o1[0], out_lo=o1[1], variant=2)
    assert torch.equal(o0[0].view(torch.int16), o1[0].view(torch.int16)) and torch.equal(o0[1].view(torch.int16), o1[1].view(torch.int16))
    if n % 64 == 0:                                                     # SwiGLU pairs
        s0 = [torch.full((m, n // 2), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        s1 = [torch.full((m, n // 2), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        ops.gemm16_fragw(hi, lo, wf, None, n, kp, ops.EPI_SWIGLU_SPLIT, out_hi=s0[0], out_lo=s0[1], variant=0, stream_k=False)
        ops.gemm16_fragw(hi, lo, wf, None, n, kp, ops.EPI_SWIGLU_SPLIT, out_hi=s1[0], out_lo=s1[1], variant=2)
        assert torch.equal(s0[0].view(torch.int16), s1[0].view(torch.int16)) and torch.equal(s0[1].view(torch.int16), s1[1].view(torch.int16))
    # plain bf16 operands (one plane, one MFMA per tile and sub-step, 4 DMA requests per K-step): the same loop, same bit-equality
    c0 = torch.full((m, n), float("nan"), device="cuda")
    c1 = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16_fragw(hi, None, wf, b, n, kp, ops.EPI_F32, c=c0, variant=0, stream_k=False)
    ops.gemm16_fragw(hi, None, wf, b, n, kp, ops.EPI_F32, c=c1, variant=2)
    assert torch.equal(c0, c1), f"plain F32: {int((c0 != c1).sum())} elements differ"
    c0, c1 = r.clone(), r.clone()
    ops.gemm16_fragw(hi, None, wf, b, n, kp, ops.EPI_RESID, c=c0, resid=c0, variant=0, stream_k=False)
    ops.gemm16_fragw(hi, None, wf, b, n, kp, ops.EPI_RESID, c=c1, resid=c1, variant=2)
    assert torch.equal(c0, c1)
    o0 = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    o1 = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm16_fragw(hi, None, wf, b, n, kp, ops.EPI_OUT16, out_hi=o0, variant=0, stream_k=False)
    ops.gemm16_fragw(hi, None, wf, b, n, kp, ops.EPI_OUT16, out_hi=o1, variant=2)
    assert torch.equal(o0.view(torch.int16), o1.view(torch.int16))
    if n % 64 == 0:
        s0 = torch.full((m, n // 2), float("nan"), dtype=torch.bfloat16, device="cuda")
        s1 = torch.full((m, n // 2), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.gemm16_fragw(hi, None, wf, None, n, kp, ops.EPI_SWIGLU16, out_hi=s0, variant=0, stream_k=False)
        ops.gemm16_fragw(hi, None, wf, None, n, kp, ops.EPI_SWIGLU16, out_hi=s1, variant=2)
        assert torch.equal(s0.view(torch.int16), s1.view(torch.int16))
    with pytest.raises(RuntimeError):                                   # an epilogue it does not have: refused, not mis-computed
        ops.gemm16_fragw(hi, lo, wf, b, n, kp, ops.EPI_QGELU_SPLIT, out_hi=o0, out_lo=o1, variant=2)


def test_gemm_bda_declines_short_k_and_the_dispatcher_falls_back():
    """K < 192 (fewer than three K-steps: the ring's prologue) is outside gemm_bda's domain: variant 2 is refused, and the library's
    own routes (variant 0 / -1, which try the DMA loop first) fall back to gemm_bd_kernel -- same bits as the LDS-staged kernel."""
    from llark_amd import ops
    m, n, k = 300, 256, 128
    g = torch.Generator().manual_seed(5)
    a = torch.randn(m, k, generator=g)
    wb = (torch.randn(n, k, generator=g) * 0.1).bfloat16()
    hi, lo = ops.split16(a.cuda(), torch.bfloat16, kmult=64)
    wt = ops.pack_weight16(wb.cuda(), False, torch.bfloat16, kmult=64)
    wf = ops.pack_weight16_frag(wt, n)
    for l in (lo, None):
        c0 = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16(hi, l, wt, None, n, ops.EPI_F32, c=c0, variant=12)
        for v in (0, -1):
            c1 = torch.full((m, n), float("nan"), device="cuda")
            ops.gemm16_fragw(hi, l, wf, None, n, k, ops.EPI_F32, c=c1, variant=v, stream_k=False)
            assert torch.equal(c0, c1)
        with pytest.raises(RuntimeError):
            ops.gemm16_fragw(hi, l, wf, None, n, k, ops.EPI_F32, c=c1, variant=2)


@pytest.mark.parametrize("m,n,k", [(333, 450, 200), (1000, 768, 1216), (128, 64, 64), (700, 300, 4800)])
def test_gemm_fragment_major_weights_bit_identical(m, n, k):
    """The B-direct kernel (fragment-major weights streamed L2 -> VGPR) accumulates every output element in the same
    k order as the LDS-staged kernel (k16 steps ascending, hi pass then lo pass), so the two agree BIT FOR BIT:
    ragged M / N (row clamping, masked stores, zero-filled weight rows of the last 32-row tile), every epilogue."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.1).half()
    b = torch.randn(n, generator=g).cuda()
    r = torch.randn(m, n, generator=g).cuda()
    hi, lo = ops.split16(a.cuda(), torch.float16, kmult=64)
    wt = ops.pack_weight16(w.cuda(), True, torch.float16, kmult=64)
    wf = ops.pack_weight16_frag(wt, n)
    kp = wt.shape[1]
    for split in (True, False):
        l = lo if split else None
        c0 = torch.full((m, n), float("nan"), device="cuda")
        c1 = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16(hi, l, wt, b, n, ops.EPI_F32, c=c0, variant=12)
        for bd_variant in (0, 1, -1):                                  # 128x256 tiles, 128x128 tiles, library choice
            c1.fill_(float("nan"))
            ops.gemm16_fragw(hi, l, wf, b, n, kp, ops.EPI_F32, c=c1, variant=bd_variant)
            assert torch.equal(c0, c1), f"F32 split={split} variant={bd_variant}: max diff {(c0 - c1).abs().max().item():.3e}"
        c0, c1 = r.clone(), r.clone()
        ops.gemm16(hi, l, wt, b, n, ops.EPI_RESID, c=c0, resid=c0, variant=12)
        ops.gemm16_fragw(hi, l, wf, b, n, kp, ops.EPI_RESID, c=c1, resid=c1)
        assert torch.equal(c0, c1)
        o0 = [torch.zeros((m, n), dtype=torch.float16, device="cuda") for _ in range(2)]
        o1 = [torch.zeros((m, n), dtype=torch.float16, device="cuda") for _ in range(2)]
        ops.gemm16(hi, l, wt, b, n, ops.EPI_QGELU_SPLIT, out_hi=o0[0], out_lo=o0[1], variant=12)
        ops.gemm16_fragw(hi, l, wf, b, n, kp, ops.EPI_QGELU_SPLIT, out_hi=o1[0], out_lo=o1[1])
        assert torch.equal(o0[0], o1[0]) and torch.equal(o0[1], o1[1])
    # the automatic route: a weight with an attached twin goes to the B-direct kernel for M >= FRAG_MIN_ROWS
    if m >= ops.FRAG_MIN_ROWS:
        ops.attach_frag(wt, n)
        c2 = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_F32, c=c2)
        c3 = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16_fragw(hi, lo, wf, b, n, kp, ops.EPI_F32, c=c3)
        assert torch.equal(c2, c3)
        ops.detach_frag(wt)
    if n % 64 == 0:                                                     # SwiGLU pairs (bf16)
        wb = (torch.randn(n, k, generator=g) * 0.1).bfloat16()
        wts = ops.pack_weight16(wb.cuda(), False, torch.bfloat16, kmult=64)
        wfs = ops.pack_weight16_frag(wts, n)
        ab, abl = ops.split16(a.cuda(), torch.bfloat16, kmult=64)
        s0 = [torch.zeros((m, n // 2), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        s1 = [torch.zeros((m, n // 2), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        ops.gemm16(ab, abl, wts, None, n, ops.EPI_SWIGLU_SPLIT, out_hi=s0[0], out_lo=s0[1], variant=12)
        ops.gemm16_fragw(ab, abl, wfs, None, n, wts.shape[1], ops.EPI_SWIGLU_SPLIT, out_hi=s1[0], out_lo=s1[1])
        assert torch.equal(s0[0], s1[0]) and torch.equal(s0[1], s1[1])
        ops.gemm16(ab, None, wts, None, n, ops.EPI_SWIGLU16, out_hi=s0[0], variant=12)
        ops.gemm16_fragw(ab, None, wfs, None, n, wts.shape[1], ops.EPI_SWIGLU16, out_hi=s1[0])
        assert torch.equal(s0[0], s1[0])


@pytest.mark.parametrize("m,n,k,epi,split", [
    (2968, 4096, 4096, "resid", False),       # o_proj at the benchmarked prefill: 384 tiles over 512 resident workgroups
    (2968, 4096, 11008, "resid", True),       # down_proj, split operands
    (2968, 12288, 4096, "f32", True),         # qkv: 1152 tiles = 1 whole round + a 640-tile stream-K pool
    (2968, 22016, 4096, "swiglu", False),     # gate/up pairs: 2064 tiles = 3 whole rounds + 528
    (2968, 22016, 4096, "swiglu", True),
    (371, 12288, 4096, "f32", False),         # one clip: 144 tiles, every tile cut
    (1500, 4000, 2048, "resid", True),        # ragged M and N
])
def test_gemm_stream_k(m, n, k, epi, split):
    """Stream-K form of the B-direct kernel (llark_gemm16_fragw_sk): tiles shared by several workgroups are finished by a
    fixed owner that adds the other partial tiles in slot order.  Against the one-workgroup-per-tile kernel the result may
    differ only by the fp32 summation order over K; it is deterministic (run to run bit-equal), leaves its flags clean for
    the next launch, and covers every epilogue the Llama prefill uses."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g, device="cuda")
    w = (torch.randn(n, k, generator=g, device="cuda") * 0.05).bfloat16()
    r = torch.randn(m, n, generator=g, device="cuda")
    hi, lo = ops.split16(a, torch.bfloat16, kmult=64)
    l = lo if split else None
    wt = ops.pack_weight16(w, False, torch.bfloat16, kmult=64)
    wf = ops.pack_weight16_frag(wt, n)
    kp = wt.shape[1]
    scratch = ops.sk_scratch()
    assert scratch is not None
    scratch.zero_()

    def run(stream_k):
        if epi == "swiglu":
            o = [torch.zeros((m, n // 2), dtype=torch.bfloat16, device="cuda") for _ in range(2)]
            ops.gemm16_fragw(hi, l, wf, None, n, kp, ops.EPI_SWIGLU_SPLIT if split else ops.EPI_SWIGLU16, out_hi=o[0],
                             out_lo=o[1] if split else None, stream_k=stream_k)
            return o[0].float() + (o[1].float() if split else 0.0)
        c = r.clone() if epi == "resid" else torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16_fragw(hi, l, wf, None, n, kp, ops.EPI_RESID if epi == "resid" else ops.EPI_F32, c=c,
                         resid=c if epi == "resid" else None, stream_k=stream_k)
        return c

    ref = run(False)
    assert int(scratch.ne(0).sum()) == 0, "the per-tile kernel must not touch the stream-K scratch"
    auto = run(None)                                                     # library choice: stream-K only for split operands, < 1 round
    tiles = math.ceil(m / 128) * math.ceil(n / 256)
    assert (int(scratch.ne(0).sum()) > 0) == (tiles < 512 and (split or 4 * tiles <= 512 or k >= 8192))
    got = run(True)
    nflag = scratch.numel() // (128 * 256 + 1) * 128 * 256              # [resident workgroups] slabs of 128x256 fp32, then the flags
    assert int(scratch[:nflag].ne(0).sum()) > 0, "stream-K did not run (no partial tile was written)"
    assert int(scratch[nflag:].ne(0).sum()) == 0, "hand-off flags must be zero again after the launch"
    scale = float(ref.abs().max())
    tol = 2e-6 * scale * (k / 4096) ** 0.5 if epi != "swiglu" else (2.0 ** -7 if not split else 2.0 ** -14) * scale
    report_close(f"stream-K {m}x{n}x{k} {epi} split={split}", got.cpu().numpy(), ref.cpu().numpy(), tol)
    report_close("library choice", auto.cpu().numpy(), ref.cpu().numpy(), tol)
    for _ in range(3):                                                   # back-to-back launches: flags re-armed, bit-equal
        again = run(True)
        assert torch.equal(again, got), "stream-K result changed from run to run"


def test_gemm_persistent_chunk_synchronous_bit_identical():
    """M >= 16384 routes to the persistent kernel (resident workgroups walk chunks of neighbouring tiles with a
    per-XCD counter barrier in between).  Same tiles, same k order => bit-identical to one-workgroup-per-tile;
    ragged M and N, several chunks per XCD (2196 tiles over 512 resident workgroups), residual epilogue in place."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    m, n, k = 70100, 1000, 128                      # 548 x 4 tiles of 128x256
    a = torch.randn(m, k, generator=g, device="cuda")
    w = (torch.randn(k, n, generator=g, device="cuda") * 0.1).half()
    b = torch.randn(n, generator=g, device="cuda")
    hi, lo = ops.split16(a, torch.float16, kmult=64)
    wt = ops.pack_weight16(w, True, torch.float16, kmult=64)
    c12 = torch.full((m, n), float("nan"), device="cuda")
    c20 = torch.full((m, n), float("nan"), device="cuda")
    cdef = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16(hi, lo, wt, b, n, ops.EPI_F32, c=c12, variant=12)
    ops.gemm16(hi, lo, wt, b, n, ops.EPI_F32, c=c20, variant=20)
    ops.gemm16(hi, lo, wt, b, n, ops.EPI_F32, c=cdef)                      # library default for this shape
    assert torch.isfinite(c20).all()
    assert torch.equal(c12, c20) and torch.equal(c12, cdef)
    ref = a.double() @ w.double() + b.double()
    worst = ((c20.double() - ref).abs() / (a.abs().double() @ w.abs().double())).max().item()
    assert worst < 6e-7, worst
    r = torch.randn(m, n, generator=g, device="cuda")
    r12, r20 = r.clone(), r.clone()
    for _ in range(3):                                                     # back-to-back launches reuse the counter ring
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_RESID, c=r12, resid=r12, variant=12)
        ops.gemm16(hi, lo, wt, b, n, ops.EPI_RESID, c=r20, resid=r20, variant=20)
    assert torch.equal(r12, r20)


def test_gemm_epilogues():
    from llark_amd import ops
    g = torch.Generator().manual_seed(9)
    m, n, k = 300, 192, 160
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.1).half()
    b = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g)
    hi, lo = ops.split16(a.cuda(), torch.float16)
    wt = ops.pack_weight16(w.cuda(), True, torch.float16)
    ref = (a.double() @ w.double() + b.double())
    # residual, in place
    c = r.clone().cuda()
    ops.gemm16(hi, lo, wt, b.cuda(), n, ops.EPI_RESID, c=c, resid=c)
    report_close("resid", c.cpu(), (r.double() + ref), 2e-5)
    # quick-gelu split output
    ohi = torch.zeros((m, n), dtype=torch.float16, device="cuda")
    olo = torch.zeros_like(ohi)
    ops.gemm16(hi, lo, wt, b.cuda(), n, ops.EPI_QGELU_SPLIT, out_hi=ohi, out_lo=olo)
    gref = ref * torch.sigmoid(1.702 * ref)
    report_close("qgelu", (ohi.float() + olo.float()).cpu(), gref, 3e-6, 3e-6)
    # plain split output
    ops.gemm16(hi, lo, wt, b.cuda(), n, ops.EPI_SPLIT16, out_hi=ohi, out_lo=olo)
    report_close("split16", (ohi.float() + olo.float()).cpu(), ref, 3e-6, 3e-6)
    # bf16 single pass + OUT16 and SwiGLU (Llama shapes)
    ab = a.bfloat16()
    wb = (torch.randn(n, k, generator=g) * 0.1).bfloat16()         # nn.Linear layout [n][k]
    wtb = ops.pack_weight16(wb.cuda(), False, torch.bfloat16)
    ahi, _ = ops.split16(ab.float().cuda(), torch.bfloat16, want_lo=False)
    o16 = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    ops.gemm16(ahi, None, wtb, None, n, ops.EPI_OUT16, out_hi=o16)
    refb = ab.double() @ wb.double().t()
    report_close("bf16 out16", o16.float().cpu(), refb, 1e-2, 8e-3)
    # swiglu: rows interleaved [32 gate | 32 up]
    inter = n // 2
    gate, up = wb[:inter], wb[inter:]
    packed = torch.stack([gate.view(-1, 32, k), up.view(-1, 32, k)], dim=1).reshape(n, k).contiguous()
    wts = ops.pack_weight16(packed.cuda(), False, torch.bfloat16)
    osw = torch.zeros((m, inter), dtype=torch.bfloat16, device="cuda")
    ops.gemm16(ahi, None, wts, None, n, ops.EPI_SWIGLU16, out_hi=osw)
    gg, uu = ab.double() @ gate.double().t(), ab.double() @ up.double().t()
    report_close("swiglu", osw.float().cpu(), torch.nn.functional.silu(gg) * uu, 1e-2, 1e-2)


@pytest.mark.parametrize("rows,width", [(5, 192), (64, 4800), (3, 1024), (9, 64)])
def test_layernorm_split(rows, width):
    from llark_amd import ops
    g = torch.Generator().manual_seed(width)
    x = torch.randn(rows, width, generator=g) * 2 + 0.5
    gam = 1 + 0.1 * torch.randn(width, generator=g)
    bet = 0.1 * torch.randn(width, generator=g)
    ld = ops.round_up(width, 32)
    hi = torch.zeros((rows, ld), dtype=torch.float16, device="cuda")
    lo = torch.zeros_like(hi)
    ops.layernorm_split(x.cuda(), gam.cuda(), bet.cuda(), 1e-5, hi, lo)
    ref = torch.nn.functional.layer_norm(x.double(), (width,), gam.double(), bet.double(), 1e-5)
    report_close("layernorm", (hi.float() + lo.float())[:, :width].cpu(), ref, 2e-6, 2e-6)


@pytest.mark.parametrize("pattern", [1, 2, 3])
@pytest.mark.parametrize("cfg", ["tiny", "full"])
def test_factored_attention(pattern, cfg):
    """vs oracle factored_attention (fp32).  Round 4: both products are three-pass split-fp16 MFMAs (q, k, v, p carried as fp16 hi + lo,
    the lo x lo term dropped): per score 2^-22 x sum|q||k| x scale ~ 4e-6 worst case at head_dim 150 with |q|, |k| ~ 1.5, typically a
    tenth of it (measured: mean 2.3e-7, max 7.3e-6 over 9.8 M outputs).  Bound 1e-5 abs + rel (was 6e-6 with exact fp32 MFMAs)."""
    from llark_amd import ops
    from oracle import jukebox_ref as R
    if cfg == "tiny":
        n, t, heads, S, blocks = 2, 512, 2, 48, 8
    else:
        n, t, heads, S, blocks = 1, 8192, 8, 1200, 128
    g = torch.Generator().manual_seed(pattern)
    qkv = torch.randn(n, t, 3 * S, generator=g)
    qkv[..., : 2 * S] *= 1.5
    q, k, v = qkv.chunk(3, dim=2)
    ref = R.factored_attention(q.contiguous(), k.contiguous(), v.contiguous(), pattern, heads, t // blocks)
    Sp = ops.round_up(S, 32)
    hi = torch.zeros((n * t, Sp), dtype=torch.float16, device="cuda")
    lo = torch.zeros_like(hi)
    ops.prior_attn(qkv.view(n * t, 3 * S).cuda(), n, t, S, heads, blocks, pattern, hi, lo)
    got = (hi.float() + lo.float())[:, :S].cpu().view(n, t, S)
    report_close(f"attn pattern {pattern} {cfg}", got, ref, 1e-5, 1e-5)
    assert (hi[:, S:] == 0).all()
    if pattern == 3:
        assert (got[:, : t // blocks] == 0).all()


def test_embed_and_pool():
    from llark_amd import ops
    from oracle import jukebox_ref as R
    hps = hparams_tiny()
    w = make_prior_weights(hps, 1)
    x_cond, y_cond = R.get_cond(w, hps)
    z = torch.randint(0, hps.l_bins, (3, hps.n_ctx), generator=torch.Generator().manual_seed(2))
    ref = R.prior_embed(w, z, x_cond, y_cond, hps)
    got = ops.prior_embed(z.cuda(), w["prior.x_emb.weight"].cuda(), w["prior.pos_emb.pos_emb"].cuda(),
                          x_cond[0].contiguous().cuda(), y_cond.reshape(-1).cuda())
    assert torch.equal(got.cpu(), ref), "embedding head must be bit-exact (same association)"
    acts = torch.randn(2, 512, 192, generator=torch.Generator().manual_seed(3))
    pw = ops.pool_window(acts.cuda(), 34, 512 // 34).cpu()
    report_close("pool_window", pw[0], R.windowed_average(acts[0], 34)[0], 1e-6)
    pm = ops.pool_mean(acts.cuda(), torch.tensor([512, 100], dtype=torch.int32).cuda()).cpu()
    report_close("pool_mean", pm[1], acts[1, :100].mean(0), 1e-6)


def _run_prior(hps, depth, n, seed=0, tap_tol=2e-5, precision="f16x2"):
    from llark_amd.jukebox.prior import TopPrior
    from llark_amd.jukebox import extract as E
    from oracle import jukebox_ref as R
    w = make_prior_weights(hps, seed + 1, depth=depth)
    z = torch.randint(0, hps.l_bins, (n, hps.n_ctx), generator=torch.Generator().manual_seed(seed))
    x_cond_r, y_cond_r = R.get_cond(w, hps)
    tp = TopPrior(hps, w, "cuda", depth=depth, precision=precision)
    x_cond, y_cond = E.get_cond(hps, tp)
    assert torch.equal(x_cond.cpu(), x_cond_r) and torch.equal(y_cond.cpu(), y_cond_r), "conditioning tables differ"
    # layer-by-layer taps (each HIP layer is fed the ORACLE's input so errors do not compound)
    h_ref = R.prior_embed(w, z, x_cond_r, y_cond_r, hps)
    tp.prior.only_encode = True
    for d in range(depth):
        taps_ref, taps = {}, {}
        h_in = h_ref.clone()
        h_ref = R.prior_layer(w, h_ref, d, hps, taps_ref)
        h_dev = h_in.cuda().view(n * hps.n_ctx, hps.prior_width).contiguous()
        tp.prior.layer_forward(h_dev, d, n, taps)
        for name in ("ln0", "qkv", "att", "xa", "ln1", "g"):
            refv = taps_ref[name].reshape(n * hps.n_ctx, -1)
            scale = refv.abs().max().item()
            report_close(f"layer {d} tap {name}", taps[name][:, : refv.shape[1]].cpu(), refv, tap_tol * scale)
        report_close(f"layer {d} out", h_dev.cpu(), h_ref.view(n * hps.n_ctx, -1), tap_tol * h_ref.abs().max().item())
    # end to end through the public entry point
    acts = E.get_final_activations(z.cuda(), x_cond, y_cond, tp)
    scale = h_ref.abs().max().item()
    worst = report_close("prior end-to-end", acts.cpu(), h_ref, 1e-4 * scale)
    return worst / scale


def test_prior_tiny_layers_and_e2e():
    rel = _run_prior(hparams_tiny(), 3, 2)
    print(f"tiny prior rel err {rel:.3e}")


def test_prior_full_width_depth3():
    """5b widths (4800 / 8 heads x 150 / 8192 tokens), 3 layers = all three attention patterns."""
    rel = _run_prior(hparams_5b_depth(3), 3, 1)
    print(f"full-width prior (3 layers) rel err {rel:.3e}")


def test_prior_folded_layernorm_matches_unfolded_and_oracle():
    """The 5b-width prior, 3 layers (all attention patterns), with the LayerNorms folded into the products (the default) against
    (a) the CPU oracle at 1e-4 of max|h|, (b) the same engine with LLARK_PRIOR_LN_FOLD off -- two roundings of one graph -- and
    (c) itself at batch 2: row statistics come from fixed-order partial sums, so clip 0 of a batch is BIT-equal to clip 0 alone."""
    from llark_amd.jukebox.prior import TopPrior
    from llark_amd.jukebox import extract as E
    from oracle import jukebox_ref as R
    hps = hparams_5b_depth(3)
    w = make_prior_weights(hps, 5, depth=3)
    z = torch.randint(0, hps.l_bins, (2, hps.n_ctx), generator=torch.Generator().manual_seed(11))
    tp_f = TopPrior(hps, w, "cuda", depth=3, ln_fold=True)
    tp_u = TopPrior(hps, w, "cuda", depth=3, ln_fold=False)
    x_cond, y_cond = E.get_cond(hps, tp_f)
    a_f1 = E.get_final_activations(z[:1].cuda(), x_cond, y_cond, tp_f)
    assert tp_f.prior._fold_rows and not tp_u.prior.ln_fold, "the folded path was not taken at 8192 rows"
    a_u1 = E.get_final_activations(z[:1].cuda(), x_cond, y_cond, tp_u)
    a_f2 = E.get_final_activations(z.cuda(), x_cond, y_cond, tp_f)
    assert torch.equal(a_f2[0], a_f1[0]), "folded LayerNorm: clip 0 of a batch of 2 differs from clip 0 alone"
    x_cond_r, y_cond_r = R.get_cond(w, hps)
    h = R.prior_embed(w, z[:1], x_cond_r, y_cond_r, hps)
    for d in range(3):
        h = R.prior_layer(w, h, d, hps)
    scale = float(h.abs().max())
    e_f = report_close("folded prior vs oracle", a_f1.cpu(), h, 1e-4 * scale)
    e_u = report_close("unfolded prior vs oracle", a_u1.cpu(), h, 1e-4 * scale)
    e_fu = float((a_f1 - a_u1).abs().max())
    print(f"\n[ln-fold] 3 layers at 5b widths: folded vs oracle {e_f:.2e}, unfolded vs oracle {e_u:.2e}, folded vs unfolded {e_fu:.2e}, max|h| {scale:.2f}")
    assert e_fu <= 2e-5 * scale


def test_prior_cproj_on_the_dma_loop_knob(monkeypatch):
    """LLARK_PRIOR_CPROJ_BDA=1 (round 5, opt-in: measured slower): the attention-output product of every block as LayerNorm producer on
    gemm_bda's tiles (llark_gemm16_lnp_fragw, 64-column partial sums) instead of the persistent tile -- same graph, another summation
    order: equal to the default folded path to rounding, inside the oracle bar, clip 0 of a batch bit-equal to clip 0 alone."""
    from llark_amd.jukebox.prior import TopPrior
    from llark_amd.jukebox import extract as E
    hps = hparams_5b_depth(3)
    w = make_prior_weights(hps, 5, depth=3)
    z = torch.randint(0, hps.l_bins, (2, hps.n_ctx), generator=torch.Generator().manual_seed(11))
    tp_d = TopPrior(hps, w, "cuda", depth=3, ln_fold=True)
    monkeypatch.setenv("LLARK_PRIOR_CPROJ_BDA", "1")
    tp_b = TopPrior(hps, w, "cuda", depth=3, ln_fold=True)
    assert tp_b.prior.cproj_bda and not tp_d.prior.cproj_bda
    x_cond, y_cond = E.get_cond(hps, tp_d)
    a_d = E.get_final_activations(z[:1].cuda(), x_cond, y_cond, tp_d)
    a_b = E.get_final_activations(z[:1].cuda(), x_cond, y_cond, tp_b)
    a_b2 = E.get_final_activations(z.cuda(), x_cond, y_cond, tp_b)
    assert tp_b.prior._fold_rows and tp_b.prior.layers[0].wf_proj is not None, "the knob's path was not taken"
    assert torch.equal(a_b2[0], a_b[0]), "clip 0 of a batch of 2 differs from clip 0 alone"
    scale = float(a_d.abs().max())
    e = float((a_b - a_d).abs().max())
    print(f"\n[cproj-bda] 3 layers at 5b widths: knob vs default folded path {e:.2e}, max|h| {scale:.2f}")
    assert e <= 2e-5 * scale


def test_prior_rejects_unsupported():
    from llark_amd.jukebox.prior import TopPrior
    hps = hparams_tiny()
    tp = TopPrior(hps, make_prior_weights(hps, 1, depth=1), "cuda", depth=1)
    z = torch.zeros((1, hps.n_ctx), dtype=torch.int64, device="cuda")
    with pytest.raises(NotImplementedError):
        tp.prior.forward(z, x_cond=None, y_cond=None)          # only_encode not set
    tp.prior.only_encode = True
    with pytest.raises(NotImplementedError):
        tp.prior.forward(z, x_cond=torch.zeros(1), y_cond=torch.zeros(1), fp16=True)


@pytest.mark.parametrize("m", [1, 8, 16])
def test_gemm_skinny_decode_shapes(m):
    """M <= 16 goes to the HBM-bound skinny kernel (decode): F32, RESID, SwiGLU (plain and split), both modes."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(100 + m)
    n, k = 448, 800                                  # n: 7 SwiGLU groups of 64; k: 25 k-steps (uneven 4-way split)
    a = torch.randn(m, k, generator=g)
    wb = (torch.randn(n, k, generator=g) * 0.1).bfloat16()
    bias = torch.randn(n, generator=g)
    wt = ops.pack_weight16(wb.cuda(), False, torch.bfloat16)
    hi, lo = ops.split16(a.cuda(), torch.bfloat16)
    a16 = hi.float().cpu()[:, :k] + lo.float().cpu()[:, :k]
    ref = a16.double() @ wb.double().t() + bias.double()
    c = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16(hi, lo, wt, bias.cuda(), n, ops.EPI_F32, c=c)
    report_close("skinny split f32", c.cpu(), ref, 2e-5, 2e-5)
    c1 = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16(hi, None, wt, None, n, ops.EPI_F32, c=c1)
    report_close("skinny plain f32", c1.cpu(), hi.float().cpu()[:, :k].double() @ wb.double().t(), 2e-5, 2e-5)
    r = torch.randn(m, n, generator=g)
    cr = r.clone().cuda()
    ops.gemm16(hi, lo, wt, bias.cuda(), n, ops.EPI_RESID, c=cr, resid=cr)
    report_close("skinny resid", cr.cpu(), r.double() + ref, 2e-5, 2e-5)
    inter = n // 2
    gate, up = wb[:inter], wb[inter:]
    packed = torch.stack([gate.view(-1, 32, k), up.view(-1, 32, k)], dim=1).reshape(n, k).contiguous()
    wts = ops.pack_weight16(packed.cuda(), False, torch.bfloat16)
    oh = torch.zeros((m, inter), dtype=torch.bfloat16, device="cuda")
    ol = torch.zeros_like(oh)
    ops.gemm16(hi, lo, wts, None, n, ops.EPI_SWIGLU_SPLIT, out_hi=oh, out_lo=ol)
    sref = torch.nn.functional.silu(a16.double() @ gate.double().t()) * (a16.double() @ up.double().t())
    report_close("skinny swiglu split", (oh.float() + ol.float()).cpu(), sref, 3e-5, 3e-5)
    ops.gemm16(hi, None, wts, None, n, ops.EPI_SWIGLU16, out_hi=oh)
    h16 = hi.float().cpu()[:, :k].double()
    report_close("skinny swiglu16", oh.float().cpu(), torch.nn.functional.silu(h16 @ gate.double().t()) * (h16 @ up.double().t()), 1e-2, 1e-2)
