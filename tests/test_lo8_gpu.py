"""The "lo8" form of the prior's split GEMM (csrc/gemm256_lo8.hip) and its producers: fp16 hi plane + E4M3 low plane in
MFMA slot order.  The checker is a float64 evaluation of exactly what the kernel is specified to compute:

    C = hi @ W^T + 2^-(sa+sw) * e4m3(lo 2^sa) @ e4m3(W 2^sw)^T (+ bias)

so what is measured here is layout, scale and accumulation -- the accuracy of the scheme itself (15-16 significant bits)
is measured end to end in tests/test_fulldepth_gpu.py and tests/test_prior_gpu.py against the exact-fp32 oracle.
"""
import numpy as np
import pytest
import torch

from conftest import report_close
from llark_amd import ops

pytestmark = pytest.mark.gpu
SA = ops.LO8_SA


def slot_order(ld):
    k = torch.arange(ld)
    r = k & 63
    return (k & ~63) + (((r >> 3) & 1) << 5) + ((r >> 4) << 3) + (r & 7)


def e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)


def make_lo8_plane(lo: torch.Tensor, kp: int) -> torch.Tensor:
    """fp32 residual [m][k] -> uint8 plane [m][kp] in MFMA slot order (what the device producers write)."""
    m, k = lo.shape
    q = torch.zeros((m, kp), dtype=torch.uint8)
    vals = e4m3(lo * 2.0 ** SA).view(torch.uint8)
    q[:, slot_order(kp)[:k]] = vals
    return q


def operands(m, n, k, seed=0, scale_a=1.0, scale_w=0.02):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(m, k, generator=g) * scale_a
    w = (torch.randn(n, k, generator=g) * scale_w).half()
    kp = ops.round_up(k, 64)
    hi = torch.zeros(m, kp, dtype=torch.float16)
    hi[:, :k] = a.half()
    lo = a - hi[:, :k].float()
    lo8 = make_lo8_plane(lo, kp)
    wt = torch.zeros(n, kp, dtype=torch.float16)
    wt[:, :k] = w
    sw = ops.lo8_weight_exponent(w)
    lo_dq = e4m3(lo * 2.0 ** SA).double() * 2.0 ** -SA
    w_dq = e4m3(w.float() * 2.0 ** sw).double() * 2.0 ** -sw
    ref = hi[:, :k].double() @ w.double().t() + lo_dq @ w_dq.t()
    exact = a.double() @ w.double().t()
    mag = a.abs().double() @ w.abs().double().t()
    return a, hi, lo8, wt, sw, ref, exact, mag


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (1000, 3600, 1216), (2048, 4800, 4800), (515, 290, 200), (8192, 1200, 640), (300, 4800, 1200),
                                   (4096, 512, 192), (700, 700, 448)])
def test_gemm_lo8_matches_its_specification(m, n, k):
    a, hi, lo8, wt, sw, ref, exact, mag = operands(m, n, k, seed=m + n + k)
    bias = torch.randn(n, generator=torch.Generator().manual_seed(1))
    c = torch.full((m, n), float("nan"), device="cuda")
    wt_d = wt.cuda()
    w8 = ops.pack_weight_lo8(wt_d, sw)
    # the packed plane is the slot-ordered E4M3 rounding of W 2^sw
    want8 = torch.zeros_like(wt, dtype=torch.uint8)
    want8[:, slot_order(wt.shape[1])] = e4m3(wt.float() * 2.0 ** sw).view(torch.uint8)
    assert torch.equal(w8.cpu(), want8)
    ops.gemm16_lo8(hi.cuda(), lo8.cuda(), wt_d, sw, bias.cuda(), n, ops.EPI_F32, c=c, w8=w8)
    torch.cuda.synchronize()
    got = c.cpu().double()
    want = ref + bias.double()
    # fp32 accumulation over k plus the MX unit's internal alignment of the 64 fp8 products (probe: ~1e-4 of the low term)
    tol = 3e-7 * mag + 1e-6
    err = (got - want).abs()
    assert torch.isfinite(got).all()
    assert bool((err <= tol).all()), f"max err {err.max():.3e} (tol {tol.max():.3e}); worst ratio {(err / tol).max():.2f}"
    # and the scheme itself: 15+ significant bits of the activation
    rel = ((got - exact - bias.double()).abs() / mag).max().item()
    assert rel < 2.0 ** -14, f"lo8 product error {rel:.3e} of sum|a||w|"
    print(f"lo8 gemm {m}x{n}x{k}: vs specification {err.max():.2e}, vs exact {rel:.2e} of sum|a||w| (sw = {sw})")


def test_gemm_lo8_residual_and_repeat_launches_share_a_workspace():
    """EPI_RESID in place (h += a.W) and many back-to-back launches on one stream: the chunk counters of the shared
    workspace are monotonic (no memset between launches) -- results must not depend on the launch index."""
    m, n, k = 4096, 4800, 1216
    a, hi, lo8, wt, sw, ref, exact, mag = operands(m, n, k, seed=5)
    h0 = torch.randn(m, n, generator=torch.Generator().manual_seed(2))
    hi_d, lo_d, wt_d = hi.cuda(), lo8.cuda(), wt.cuda()
    w8 = ops.pack_weight_lo8(wt_d, sw)
    outs = []
    for it in range(6):
        h = h0.clone().cuda()
        ops.gemm16_lo8(hi_d, lo_d, wt_d, sw, None, n, ops.EPI_RESID, c=h, resid=h, w8=w8)
        outs.append(h)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "repeated launches sharing a workspace differ"
    err = (outs[0].cpu().double() - (h0.double() + ref)).abs()
    assert bool((err <= 3e-7 * mag + 3e-7 * h0.abs().double() + 1e-6).all()), f"max err {err.max():.3e}"


def test_gemm_lo8_qgelu_epilogue_writes_hi_and_e4m3_planes():
    m, n, k = 777, 4800, 4800
    a, hi, lo8, wt, sw, ref, exact, mag = operands(m, n, k, seed=9, scale_w=0.03)
    bias = torch.randn(n, generator=torch.Generator().manual_seed(3)) * 0.1
    np_ = ops.round_up(n, 64)
    out_hi = torch.zeros((m, np_), dtype=torch.float16, device="cuda")
    out_lo8 = torch.zeros((m, np_), dtype=torch.uint8, device="cuda")
    wt_d = wt.cuda()
    ops.gemm16_lo8(hi.cuda(), lo8.cuda(), wt_d, sw, bias.cuda(), n, ops.EPI_QGELU_SPLIT8, out_hi=out_hi, out_lo8=out_lo8,
                   w8=ops.pack_weight_lo8(wt_d, sw))
    torch.cuda.synchronize()
    x = ref + bias.double()
    g = x * torch.sigmoid(1.702 * x)
    got = out_hi.cpu().double()[:, :n] + ops.lo8_decode(out_lo8, n).cpu().double()
    # hi + lo8 carries >= 15 significant bits of g; g itself inherits the product's fp32 accumulation noise
    err = (got - g).abs()
    tol = 2.0 ** -15 * g.abs() + 2.0 ** -21 + 2e-6 * mag
    assert bool((err <= tol).all()), f"max err {err.max():.3e}; worst ratio {(err / tol).max():.2f}"
    # the hi plane is the fp16 rounding of g (up to the accumulation noise moving a value across a rounding boundary)
    hi_err = (out_hi.cpu().double()[:, :n] - g).abs()
    assert bool((hi_err <= 2.0 ** -11 * g.abs() + 2.0 ** -24 + 2e-6 * mag).all())


@pytest.mark.parametrize("rows,width", [(5, 192), (64, 4800), (3, 1024), (9, 64)])
def test_layernorm_split_lo8(rows, width):
    g = torch.Generator().manual_seed(rows * width)
    x = torch.randn(rows, width, generator=g) * 3 + 0.5
    gamma = 1 + 0.2 * torch.randn(width, generator=g)
    beta = 0.1 * torch.randn(width, generator=g)
    wp = ops.round_up(width, 64)
    hi = torch.zeros((rows, wp), dtype=torch.float16, device="cuda")
    lo8 = torch.zeros((rows, wp), dtype=torch.uint8, device="cuda")
    hi16 = torch.zeros((rows, wp), dtype=torch.float16, device="cuda")
    lo16 = torch.zeros((rows, wp), dtype=torch.float16, device="cuda")
    ops.layernorm_split_lo8(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, hi, lo8)
    ops.layernorm_split(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, hi16, lo16)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (width,), gamma.double(), beta.double(), 1e-5)
    # the two forms assign columns to lanes differently (mean / variance sums in another order): hi planes agree to one fp16 ulp
    hd = (hi.cpu().double() - hi16.cpu().double()).abs()[:, :width]
    assert bool((hd <= 2.0 ** -10 * ref.abs() + 2.0 ** -24).all()) and float((hd > 0).double().mean()) < 1e-3
    # the e4m3 plane is the round-to-nearest (3 mantissa bits) of (y - hi) at 2^12, in MFMA slot order
    got = ops.lo8_decode(lo8, width).cpu().double()
    resid = ref - hi.cpu().double()[:, :width]                    # what the low plane has to carry (up to the kernel's fp32 rounding of y)
    err = (got - resid).abs()
    tol = 2.0 ** -4 * resid.abs() + 2.0 ** -9 * 2.0 ** -SA + 3e-7 * ref.abs() + 1e-7
    assert bool((err <= tol).all()), f"max err {err.max():.3e}; worst ratio {(err / tol).max():.2f}"
    report_close("hi + lo8 vs LayerNorm", hi.cpu().double()[:, :width] + got, ref, 2.0 ** -15 * ref.abs().max().item() + 1e-6)
    if wp > width:
        assert int(hi.cpu()[:, width:].abs().sum()) == 0           # pad columns of the fp16 plane stay zero


def test_prior_tiny_and_full_width_depth3_lo8():
    """Layer taps and end-to-end activations of the lo8 prior against the exact-fp32 oracle."""
    import test_prior_gpu as TP
    from llark_amd.jukebox.hparams import hparams_tiny

    rel = TP._run_prior(hparams_tiny(), 3, 2, tap_tol=1e-4, precision="lo8")
    print(f"tiny prior (lo8) rel err {rel:.3e}")
    rel = TP._run_prior(TP.hparams_5b_depth(3), 3, 1, tap_tol=1e-4, precision="lo8")
    print(f"full-width prior, 3 layers (lo8) rel err {rel:.3e}")


def test_gemm_lo8_row_chunks_when_the_a_plane_outgrows_32bit_offsets(monkeypatch):
    """llark_gemm16_lo8 addresses A with 32-bit byte offsets (about 28 clips of the 5b prior); ops.gemm16_lo8 then
    issues row ranges of whole tiles.  Forced here with a tiny limit: bit-identical to the single launch."""
    assert ops.lo8_max_rows(4800, 4800) == ((1 << 31) - 1) // 9600 // 256 * 256 and ops.lo8_max_rows(4800, 4800) * 9600 < 1 << 31
    m, n, k = 1000, 520, 448
    a, hi, lo8, wt, sw, ref, exact, mag = operands(m, n, k, seed=77)
    hi_d, lo_d, wt_d = hi.cuda(), lo8.cuda(), wt.cuda()
    w8 = ops.pack_weight_lo8(wt_d, sw)
    h0 = torch.randn(m, n, generator=torch.Generator().manual_seed(4)).cuda()
    one = h0.clone()
    ops.gemm16_lo8(hi_d, lo_d, wt_d, sw, None, n, ops.EPI_RESID, c=one, resid=one, w8=w8)
    monkeypatch.setattr(ops, "lo8_max_rows", lambda lda, lda8: 256)
    many = h0.clone()
    ops.gemm16_lo8(hi_d, lo_d, wt_d, sw, None, n, ops.EPI_RESID, c=many, resid=many, w8=w8)
    torch.cuda.synchronize()
    assert torch.equal(one, many)
    with pytest.raises(ValueError, match="w8"):
        ops.gemm16_lo8(hi_d, lo_d, wt_d, sw, None, n, ops.EPI_RESID, c=many, resid=many)
