"""GPU parity: llark_gemm16_t (csrc/gemm_tn.hip) -- the backward products of nn.Linear on operands stored contraction-major, read
through ds_read_b64_tr_b16 -- vs a plain torch fp32 product of the same 16-bit operands (what autograd computes for
dX = dY W and dW = dY^T X under m2t/models/llamav2.py:259-337).

Tolerance: identical 16-bit operand values, fp32 accumulation on both sides in different orders: |err| <= 2e-6 * sum_k |a||w|
(a few fp32 ulps of the absolute-value product), which a swapped row / column / k-slot misses by orders of magnitude (random,
non-symmetric operands)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(c, a_mk, w_nk, init=None):
    ref = a_mk.double() @ w_nk.double().t()
    bound = a_mk.double().abs() @ w_nk.double().abs().t()
    if init is not None:
        ref = ref + init.double()
        bound = bound + init.double().abs()
    err = (c.double() - ref).abs()
    assert (err <= 2e-6 * bound + 1e-30).all(), (err / (bound + 1e-30)).max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,kp", [(128, 128, 64), (200, 136, 128), (4096, 4096, 2048), (1000, 72, 448), (8, 8, 64), (352, 12288, 192),
                                    (4096, 4096, 4096), (4100, 4360, 4096)])      # the last two: the 128 x 256 tile (K >= 4096, >= 512 tiles)
def test_dx_form_w_contraction_major(dtype, m, n, kp):
    """dX[m][n] = sum_k dY[m][k] W[k][n]: W stored [kp][n] (trans_b)."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n)
    a = torch.randn(m, kp, generator=g, device="cuda").to(dtype)
    w = (torch.randn(kp, n, generator=g, device="cuda") * 0.1).to(dtype)
    c = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16_t(a, w, m, n, kp, False, True, c)
    _check(c, a.float(), w.float().t())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,kp,accumulate", [(128, 128, 64, False), (136, 200, 128, True), (4096, 11008, 1024, True), (72, 1000, 448, False),
                                               (12288, 4096, 512, False), (12288, 4096, 4096, True), (4104, 4360, 4096, False)])   # 128 x 256 tile
def test_dw_form_both_contraction_major(dtype, m, n, kp, accumulate):
    """dW[m][n] (+)= sum_k dY[k][m] X[k][n]: both operands stored with the contraction index as the row."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(m * 5 + n)
    a = torch.randn(kp, m, generator=g, device="cuda").to(dtype)
    w = (torch.randn(kp, n, generator=g, device="cuda") * 0.1).to(dtype)
    init = torch.randn(m, n, generator=g, device="cuda") if accumulate else None
    c = init.clone() if accumulate else torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16_t(a, w, m, n, kp, True, True, c, accumulate=accumulate)
    _check(c, a.float().t(), w.float().t(), init)


def test_a_contraction_major_only_and_strided_views():
    """trans_a alone, and operands that are column slices of wider buffers (lda / ldw > free size)."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    kp, m, n = 256, 264, 136
    abuf = torch.randn(kp, m + 40, generator=g, device="cuda").bfloat16()
    wbuf = (torch.randn(n, kp + 64, generator=g, device="cuda") * 0.1).bfloat16()
    a, w = abuf[:, :m], wbuf[:, :kp]
    c = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16_t(a, w, m, n, kp, True, False, c)
    _check(c, a.float().t(), w.float())


def test_rejects_what_it_cannot_do():
    from llark_amd import ops
    from llark_amd._lib import LlarkHipError
    a = torch.zeros(64, 100, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(64, 128, dtype=torch.bfloat16, device="cuda")
    c = torch.zeros(100, 128, device="cuda")
    with pytest.raises(LlarkHipError):
        ops.gemm16_t(a, w, 100, 128, 64, True, True, c)            # m % 8 != 0 for a transposed operand
    with pytest.raises(LlarkHipError):
        ops.gemm16_t(w, w, 128, 128, 32, True, True, c)            # kp % 64 != 0


@pytest.mark.parametrize("rows,width,dtype,kmult", [(2048, 4096, torch.bfloat16, 64), (333, 200, torch.float16, 64), (7, 4800, torch.float16, 32),
                                                    (65, 1216, torch.bfloat16, 64), (5, 130, torch.bfloat16, 64), (4100, 36, torch.float16, 32)])
def test_split16_vector_and_scalar_forms(rows, width, dtype, kmult):
    """llark_split16 (fp32 -> 16-bit hi + lo planes, zero-padded to the K multiple): the four-columns-per-thread form (width % 4 == 0)
    and the scalar form (width = 130) give exactly torch's rounding; more than 4096 rows exercise the row loop."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(rows + width)
    x = torch.randn(rows, width, generator=g, device="cuda") * 3
    hi, lo = ops.split16(x, dtype, want_lo=True, kmult=kmult)
    ref_hi = x.to(dtype)
    ref_lo = (x - ref_hi.float()).to(dtype)
    assert hi.shape[1] % kmult == 0
    assert torch.equal(hi[:, :width], ref_hi) and torch.equal(lo[:, :width], ref_lo)
    if hi.shape[1] > width:
        assert hi[:, width:].abs().max().item() == 0 and lo[:, width:].abs().max().item() == 0


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("form", ["dw", "dx"])
def test_two_stage_variants_bit_identical_to_the_single_stage_kernel(variant, form):
    """Round 6: the double-buffered LDS-DMA forms (128x256x32 / 128x256x64 / 256x256x64 tiles, llark_gemm16_t_ex) accumulate every output
    element in the same order (k ascending, one 32x32x16 MFMA per 16 k) as the single-stage kernel: equal bits, on whole and ragged
    tiles, with and without accumulation into c, and with the gradient-norm side output."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(31 + variant)
    for (m, n, kp, accumulate) in [(4096, 1024, 4096, True), (1000, 520, 448, False), (264, 8, 64, True), (12288, 4096, 4096, False)]:
        if form == "dw":
            a = torch.randn(kp, m, generator=g, device="cuda").bfloat16()
            ta = True
        else:
            a = torch.randn(m, kp, generator=g, device="cuda").bfloat16()
            ta = False
        w = (torch.randn(kp, n, generator=g, device="cuda") * 0.1).bfloat16()
        init = torch.randn(m, n, generator=g, device="cuda")
        outs, sums = [], []
        for v in (0, variant):
            c = init.clone() if accumulate else torch.full((m, n), float("nan"), device="cuda")
            ss = torch.zeros(1, dtype=torch.float64, device="cuda")
            ops.gemm16_t(a, w, m, n, kp, ta, True, c, accumulate=accumulate, sumsq=ss, variant=v)
            outs.append(c)
            sums.append(ss.item())
        assert torch.equal(outs[0], outs[1]), (variant, form, m, n, kp, (outs[0] - outs[1]).abs().max().item())
        _check(outs[1], a.float().t() if ta else a.float(), w.float().t(), init if accumulate else None)
        assert abs(sums[0] - sums[1]) <= 1e-6 * abs(sums[0])            # (per-wave partials meet in a double atomic: order varies)


@pytest.mark.parametrize("kp,n", [(64, 8), (128, 200), (4096, 4096), (192, 11008), (448, 72)])
def test_pack_frag_t16_equals_transpose_then_pack(kp, n):
    """llark_pack_frag_t16 (round 6) = llark_pack_weight16_frag of the transposed operand, without the intermediate transpose; also from a
    column slice of a wider buffer."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(kp + n)
    buf = torch.randn(kp, n + 24, generator=g, device="cuda").bfloat16()
    x = buf[:, :n]
    got = ops.pack_frag_t16(x, n)
    xt = torch.zeros(ops.round_up(n, 32), kp, dtype=torch.bfloat16, device="cuda")
    xt[:n] = x.t()
    assert torch.equal(got, ops.pack_weight16_frag(xt, ops.round_up(n, 32)))


@pytest.mark.parametrize("m,n,kp,accumulate", [(128, 256, 192, False), (136, 200, 256, True), (4096, 11008, 1024, True), (72, 1000, 448, False),
                                               (12288, 4096, 4096, True), (4104, 4360, 4096, False), (22016, 4096, 4096, True)])
def test_dw_form_on_the_dma_loop(m, n, kp, accumulate):
    """llark_gemm16_ta_fragw (round 6): dW[m][n] (+)= sum_k dY[k][m] X[k][n] with dY contraction-major through the transposing LDS read of the
    DMA loop and X^T fragment-major; same bound as the llark_gemm16_t tests, and EQUAL BITS to llark_gemm16_t (same MFMA, k ascending per
    accumulator); sum-of-squares side output against the stored values."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(m * 3 + n)
    a = torch.randn(kp, m, generator=g, device="cuda").bfloat16()
    x = (torch.randn(kp, n, generator=g, device="cuda") * 0.1).bfloat16()
    init = torch.randn(m, n, generator=g, device="cuda") if accumulate else None
    c = init.clone() if accumulate else torch.full((m, n), float("nan"), device="cuda")
    ss = torch.zeros(1, dtype=torch.float64, device="cuda")
    ops.gemm16_ta_fragw(a, ops.pack_frag_t16(x, n), m, n, kp, c, accumulate=accumulate, sumsq=ss)
    _check(c, a.float().t(), x.float().t(), init)
    c2 = init.clone() if accumulate else torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16_t(a, x, m, n, kp, True, True, c2, accumulate=accumulate)
    assert torch.equal(c, c2)
    ref_ss = (c.double() ** 2).sum().item()
    assert abs(ss.item() - ref_ss) <= 1e-5 * ref_ss


@pytest.mark.parametrize("m,n,kp,accumulate", [(128, 256, 256, False), (136, 200, 384, True), (4096, 11008, 1024, True), (72, 1000, 512, False),
                                               (12288, 4096, 4096, True), (4104, 4360, 4096, False), (22016, 4096, 4096, True)])
def test_dw_form_on_the_16x16x32_mfma_shape(m, n, kp, accumulate):
    """llark_gemm16_ta_fragw16 over llark_pack_frag_t16x16 (csrc/gemm_bda16.hip): the dW product with 32 products per MFMA.  Same bound
    against the fp64 product as every kernel of this file; against llark_gemm16_ta_fragw a few fp32 ulps of the absolute-value product
    (different grouping of the same products); sum-of-squares side output against the stored values; reproducible bit for bit."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(m * 3 + n)
    a = torch.randn(kp, m, generator=g, device="cuda").bfloat16()
    x = (torch.randn(kp, n, generator=g, device="cuda") * 0.1).bfloat16()
    init = torch.randn(m, n, generator=g, device="cuda") if accumulate else None
    assert ops.gemm16_ta_fragw16_takes(m, n, kp, a.stride(0), init if accumulate else torch.empty(m, n, device="cuda"))
    outs = []
    for _ in range(2):
        c = init.clone() if accumulate else torch.full((m, n), float("nan"), device="cuda")
        ss = torch.zeros(1, dtype=torch.float64, device="cuda")
        ops.gemm16_ta_fragw(a, ops.pack_frag_t16(x, n, chunk16=True), m, n, kp, c, accumulate=accumulate, sumsq=ss, chunk16=True)
        outs.append(c)
    c = outs[0]
    assert torch.equal(outs[0], outs[1])
    _check(c, a.float().t(), x.float().t(), init)
    ref_ss = (c.double() ** 2).sum().item()
    assert abs(ss.item() - ref_ss) <= 1e-5 * ref_ss
    c2 = init.clone() if accumulate else torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16_ta_fragw(a, ops.pack_frag_t16(x, n), m, n, kp, c2, accumulate=accumulate)
    bound = a.float().t().abs() @ x.float().abs() + (init.abs() if accumulate else 0)
    assert ((c - c2).abs() <= 4e-6 * bound + 1e-30).all()


def test_dw_16x16x32_rejects_what_it_cannot_take():
    from llark_amd import ops
    from llark_amd._lib import LlarkHipError
    a = torch.zeros(192, 128, dtype=torch.bfloat16, device="cuda")
    x = torch.zeros(192, 256, dtype=torch.bfloat16, device="cuda")
    c = torch.zeros(128, 256, device="cuda")
    assert not ops.gemm16_ta_fragw16_takes(128, 256, 192, 128, c)
    with pytest.raises(LlarkHipError):
        ops.gemm16_ta_fragw(a, ops.pack_frag_t16(x, 256, chunk16=True), 128, 256, 192, c, chunk16=True)      # kp % 128 != 0
