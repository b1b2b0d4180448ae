"""GPU parity: llark_gemv16_dma (csrc/gemv_dma.hip), the weight-streaming decode-step Linear, vs a torch fp64 product of the same
bf16 operands -- what nn.Linear computes for one new token under LlamaModel.forward (m2t/infer.py:146 -> model.generate).

Tolerance: bf16 x bf16 products are exact in fp32; the kernel adds them in fp32 in its own order (per-lane dot2 chains, wave
reduction, fixed-order sum over 8 waves): |err| <= 4e-6 * sum_k |a||w| + the residual's own rounding.  SwiGLU output is bf16:
one bf16 ulp on top."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _always_stream(monkeypatch):
    """ops.gemm16 sends only weights >= 64 MB to the streaming kernel (below that the MFMA skinny kernel's smaller fixed cost wins);
    these tests want the streaming kernel for every shape."""
    from llark_amd import ops
    monkeypatch.setattr(ops, "GEMV_DMA_MIN_BYTES", 0)


def _planes(a):
    hi = a.bfloat16()
    lo = (a - hi.float()).bfloat16()
    return hi, lo


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("m,n,k", [(1, 12288, 4096), (1, 4096, 11008), (2, 4096, 4096), (3, 640, 256), (4, 32004, 4096), (1, 100, 4104), (1, 16, 12288)])
def test_f32_and_residual(split, m, n, k):
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(m * 100 + n)
    a = torch.randn(m, k, generator=g, device="cuda")
    w = (torch.randn(n, k, generator=g, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(n, generator=g, device="cuda")
    hi, lo = _planes(a)
    aeff = (hi.float() + lo.float()) if split else hi.float()
    ref = aeff.double() @ w.double().t()
    bound = aeff.double().abs() @ w.double().abs().t()
    c = torch.full((m, n), float("nan"), device="cuda")
    ops.gemm16(hi, lo if split else None, w, bias, n, ops.EPI_F32, c=c)
    assert ((c.double() - (ref + bias.double())).abs() <= 4e-6 * bound + 1e-6 * bias.abs().double()).all()
    r0 = torch.randn(m, n, generator=g, device="cuda")
    c2 = r0.clone()
    ops.gemm16(hi, lo if split else None, w, None, n, ops.EPI_RESID, c=c2, resid=c2)           # in place, as the decode step does
    assert ((c2.double() - (ref + r0.double())).abs() <= 4e-6 * bound + 2e-7 * (ref.abs() + r0.abs().double())).all()


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("m,inter,k", [(1, 11008, 4096), (2, 96, 512), (4, 1024, 4096)])
def test_swiglu(split, m, inter, k):
    """gate / up rows interleaved [gate 32 | up 32] per 64 weight rows (the packed layout of the engine)."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(inter + m)
    a = torch.randn(m, k, generator=g, device="cuda")
    gate = (torch.randn(inter, k, generator=g, device="cuda") * 0.05).bfloat16()
    up = (torch.randn(inter, k, generator=g, device="cuda") * 0.05).bfloat16()
    packed = torch.stack([gate.view(-1, 32, k), up.view(-1, 32, k)], dim=1).reshape(2 * inter, k).contiguous()
    hi, lo = _planes(a)
    aeff = (hi.float() + lo.float()) if split else hi.float()
    ref = torch.nn.functional.silu(aeff.double() @ gate.double().t()) * (aeff.double() @ up.double().t())
    oh = torch.full((m, inter), float("nan"), dtype=torch.bfloat16, device="cuda")
    ol = torch.full((m, inter), float("nan"), dtype=torch.bfloat16, device="cuda")
    if split:
        ops.gemm16(hi, lo, packed, None, 2 * inter, ops.EPI_SWIGLU_SPLIT, out_hi=oh, out_lo=ol)
        got = oh.double() + ol.double()
        tol = 2 ** -15
    else:
        ops.gemm16(hi, None, packed, None, 2 * inter, ops.EPI_SWIGLU16, out_hi=oh)
        got = oh.double()
        tol = 2 ** -8
    assert ((got - ref).abs() <= tol * ref.abs() + 1e-5).all()


def test_same_result_as_the_mfma_skinny_kernel_within_rounding():
    """The two decode kernels (this one and gemm.hip's MFMA skinny kernel, variant >= 0 forces it) agree to fp32 rounding."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(1, 4096, generator=g, device="cuda")
    w = (torch.randn(12288, 4096, generator=g, device="cuda") * 0.02).bfloat16()
    hi, lo = _planes(a)
    c1 = torch.empty(1, 12288, device="cuda")
    c2 = torch.empty(1, 12288, device="cuda")
    ops.gemm16(hi, lo, w, None, 12288, ops.EPI_F32, c=c1)
    ops.gemm16(hi, lo, w, None, 12288, ops.EPI_F32, c=c2, variant=0)
    assert (c1 - c2).abs().max().item() <= 2e-5 * c2.abs().max().item()


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("n,k,swiglu", [(12288, 4096, False), (22016, 4096, True), (640, 512, False)])
def test_fused_rmsnorm_is_bit_identical_to_the_separate_launch(split, n, k, swiglu):
    """llark_gemv16_dma_rmsnorm == llark_rmsnorm_bf16 followed by llark_gemv16_dma, bit for bit (same rstd, same hi / lo planes,
    same stream): the decode step may fuse LlamaRMSNorm into q/k/v_proj, gate/up_proj and lm_head without changing a token."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n + k)
    x = torch.randn(1, k, generator=g, device="cuda") * 3.0
    gam = 1.0 + 0.2 * torch.randn(k, generator=g, device="cuda")
    w = (torch.randn(n, k, generator=g, device="cuda") * 0.05).bfloat16()
    hi = torch.empty(1, k, dtype=torch.bfloat16, device="cuda")
    lo = torch.empty_like(hi) if split else None
    ops.rmsnorm_bf16(x, gam, 1e-5, hi, lo)
    if swiglu:
        epi = ops.EPI_SWIGLU_SPLIT if split else ops.EPI_SWIGLU16
        o1, l1, o2, l2 = (torch.full((1, n // 2), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(4))
        ops.gemm16(hi, lo, w, None, n, epi, out_hi=o1, out_lo=l1 if split else None)
        ops.gemm16_rmsnorm_a(x, gam, 1e-5, w, n, epi, split, out_hi=o2, out_lo=l2 if split else None)
        assert torch.equal(o1, o2) and (not split or torch.equal(l1, l2))
    else:
        c1 = torch.full((1, n), float("nan"), device="cuda")
        c2 = torch.full((1, n), float("nan"), device="cuda")
        ops.gemm16(hi, lo, w, None, n, ops.EPI_F32, c=c1)
        ops.gemm16_rmsnorm_a(x, gam, 1e-5, w, n, ops.EPI_F32, split, c=c2)
        assert torch.equal(c1, c2)
