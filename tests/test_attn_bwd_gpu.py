"""GPU parity: the flash-style attention backward (csrc/attn_bwd.hip) and the log-sum-exp the training forward leaves for it
vs a plain torch fp32 reference of the same op -- softmax(q k^T / sqrt(d) + causal mask) v, what LlamaAttention's eager path
computes (transformers==4.29.2 modeling_llama.py; m2t/models/llamav2.py:259-337 under loss.backward()).

Tolerance: operands are bf16 on both sides (the reference uses the same bf16-valued q, k, v, dO in fp32 arithmetic); the kernel
rounds P and dS to bf16 before their products (one rounding of 2^-9 relative per element, averaged over the contraction), so each
gradient is held to relative Frobenius error <= 1e-2 and cosine >= 0.9999; the log-sum-exp to 2e-3 absolute."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

HD = 128


def _case(B, nh, S, smax, seed):
    g = torch.Generator().manual_seed(seed)
    q = (torch.randn(B, nh, S, HD, generator=g) * 1.5).bfloat16()
    k = (torch.randn(B, nh, S, HD, generator=g) * 1.5).bfloat16()
    v = torch.randn(B, nh, S, HD, generator=g).bfloat16()
    dO = (torch.randn(B, nh, S, HD, generator=g) * 0.1).bfloat16()
    return q, k, v, dO


def _reference(q, k, v, dO, slopes=None):
    q, k, v = (t.float().cuda().requires_grad_(True) for t in (q, k, v))
    S = q.shape[2]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(HD)
    if slopes is not None:                   # ALiBi (m2t/llava/model/mpt/attention.py build_alibi_bias): slope_h * (key - (S - 1))
        s = s + slopes.cuda().view(1, -1, 1, 1) * (torch.arange(S, device="cuda").float() - (S - 1)).view(1, 1, 1, S)
    mask = torch.ones(S, S, dtype=torch.bool, device="cuda").tril()
    s = s.masked_fill(~mask, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    o = torch.softmax(s, dim=-1) @ v
    o.backward(dO.float().cuda())
    return o.detach(), lse.detach(), q.grad, k.grad, v.grad


def _run(q, k, v, dO, smax, slopes=None):
    from llark_amd import ops
    B, nh, S, _ = q.shape
    BH = B * nh
    bf = dict(dtype=torch.bfloat16, device="cuda")
    f32 = dict(dtype=torch.float32, device="cuda")
    kc = torch.zeros((B, nh, smax, HD), **bf)
    vtc = torch.zeros((B, nh, HD, smax), **bf)
    kc[:, :, :S] = k.cuda()
    vtc[:, :, :, :S] = v.cuda().transpose(-1, -2)
    qd = q.cuda().contiguous()
    att = torch.empty((B * S, nh * HD), **bf)
    lse = torch.empty((BH, S), **f32)
    sl = slopes.cuda().float().contiguous() if slopes is not None else None
    ops.attn_prefill_lse(qd, kc, vtc, B, S, nh, HD, att, lse, alibi_slopes=sl)
    dOd = dO.cuda().contiguous().view(BH, S, HD)
    v_rm = v.cuda().contiguous().view(BH, S, HD)
    dq, dk, dv = (torch.empty((BH, S, HD), **f32) for _ in range(3))
    dsum = torch.empty((BH, S), **f32)
    ops.attn_backward(qd.view(BH, S, HD), kc, v_rm, dOd, att, lse, dsum, B, S, nh, HD, dq, dk, dv, alibi_slopes=sl)
    torch.cuda.synchronize()
    o = att.view(B, S, nh, HD).permute(0, 2, 1, 3).float()
    return o, lse.view(B, nh, S), dq.view(B, nh, S, HD), dk.view(B, nh, S, HD), dv.view(B, nh, S, HD)


@pytest.mark.parametrize("B,nh,S,smax,alibi", [(1, 2, 64, 64, False), (2, 3, 100, 128, False), (1, 2, 130, 256, False), (1, 4, 1024, 1024, False),
                                               (2, 2, 333, 512, False), (1, 1, 7, 64, False), (2, 8, 300, 320, False), (2, 4, 200, 256, True),
                                               (1, 16, 515, 576, True),
                                               # round 6: grids on which the kernels take TWO causal blocks per workgroup (64 heads x 16 / 15 blocks = 512
                                               # workgroups of pairs): ragged last block, odd block count (the middle block alone), ALiBi, and the
                                               # two-query-set forward (>= 1024 blocks of 128 queries)
                                               (2, 32, 1000, 1024, False), (2, 32, 960, 1024, False), (2, 32, 960, 1024, True), (2, 32, 2000, 2048, False)])
def test_attention_backward_vs_torch_fp32(B, nh, S, smax, alibi):
    q, k, v, dO = _case(B, nh, S, smax, seed=S)
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / nh) for i in range(nh)]) if alibi else None
    ro, rlse, rdq, rdk, rdv = _reference(q, k, v, dO, slopes)
    o, lse, dq, dk, dv = _run(q, k, v, dO, smax, slopes)
    assert (lse - rlse).abs().max().item() <= 2e-3
    assert (o - ro).abs().max().item() <= 2e-2 * ro.abs().max().item()
    for name, got, ref in (("dq", dq, rdq), ("dk", dk, rdk), ("dv", dv, rdv)):
        assert torch.isfinite(got).all(), name
        rel = ((got - ref).norm() / ref.norm()).item()
        cos = torch.nn.functional.cosine_similarity(got.reshape(1, -1), ref.reshape(1, -1)).item()
        assert rel <= 1e-2 and cos >= 0.9999, (name, rel, cos)
        # per-row check: a wrong tile or mask edge moves single rows by O(1), far more than the norm shows.  The floor of 5 % of
        # the largest row norm keeps the first queries out of the ratio: there dS = P (dP - D) is a difference of nearly equal
        # numbers (one or two visible keys), so the bf16 rounding of O inside D = rowsum(dO * O) is a large RELATIVE error of a
        # negligible gradient (0.10 at S = 64 with a 0.1 % floor, reproduced by an fp64 emulation of the same roundings).
        rowrel = ((got - ref).norm(dim=-1) / (ref.norm(dim=-1) + 5e-2 * ref.norm(dim=-1).max())).max().item()
        assert rowrel <= 8e-2, (name, rowrel)


def test_attention_backward_matches_materialised_path():
    """Same gradients as the materialising kernels (scores -> llark_causal_softmax_rows -> llark_attn_ds -> three batched products)."""
    from llark_amd import ops
    B, nh, S, smax = 1, 2, 200, 256
    q, k, v, dO = _case(B, nh, S, smax, seed=3)
    _, _, dq, dk, dv = _run(q, k, v, dO, smax)
    BH, Sp = B * nh, ops.round_up(S, 64)
    bf = dict(dtype=torch.bfloat16, device="cuda")
    f32 = dict(dtype=torch.float32, device="cuda")
    qd, kd, vd, dOd = (t.cuda().contiguous().view(BH, S, HD) for t in (q, k, v, dO))
    sc = torch.empty((BH, S, S), **f32)
    ops.gemm16_batched(qd, S * HD, HD, kd, S * HD, HD, S, S, HD, BH, sc, S, S * S)
    P = torch.zeros((BH, S, Sp), **bf)
    ops.causal_softmax_rows(sc, BH, S, 1.0 / math.sqrt(HD), P)
    ops.gemm16_batched(dOd, S * HD, HD, vd, S * HD, HD, S, S, HD, BH, sc, S, S * S)
    dS = torch.zeros((BH, S, Sp), **bf)
    ops.attn_ds(P, sc, BH, S, 1.0 / math.sqrt(HD), dS)
    Pf, dSf = P[:, :, :S].float(), dS[:, :, :S].float()
    mdv = Pf.transpose(1, 2) @ dOd.float()
    mdq = dSf @ kd.float()
    mdk = dSf.transpose(1, 2) @ qd.float()
    for name, got, ref in (("dq", dq, mdq), ("dk", dk, mdk), ("dv", dv, mdv)):
        rel = ((got.view(BH, S, HD) - ref).norm() / ref.norm()).item()
        assert rel <= 5e-3, (name, rel)


@pytest.mark.parametrize("B,S,nh,smax", [(2, 200, 2, 256), (2, 2048, 32, 2048), (2, 1000, 32, 1024)])
def test_fused_glue_equals_the_separate_passes(B, S, nh, smax):
    """llark_attn_backward_bf16_fused (round 6): dO read token-major + d(q | k | v) written as bf16 with the RoPE backward in the epilogues
    == llark_split_heads16 + llark_attn_backward_bf16 + llark_rope_merge_bwd, bit for bit (same accumulators, same expressions)."""
    import torch
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(12)
    hd = 128                                                             # (the second and third shapes: the step's own, on the paired-block grids)
    H = nh * hd
    bf, f32 = torch.bfloat16, torch.float32
    q = (torch.randn(B * nh, S, hd, generator=g, device="cuda") * 0.5).to(bf)
    kc = torch.zeros(B, nh, smax, hd, dtype=bf, device="cuda")
    kc[:, :, :S] = (torch.randn(B, nh, S, hd, generator=g, device="cuda") * 0.5).to(bf)
    v_rm = (torch.randn(B * nh, S, hd, generator=g, device="cuda") * 0.5).to(bf)
    vt = torch.zeros(B, nh, hd, smax, dtype=bf, device="cuda")
    vt[:, :, :, :S] = v_rm.view(B, nh, S, hd).transpose(2, 3)
    att = torch.empty(B * S, H, dtype=bf, device="cuda")
    lse = torch.empty(B * nh, S, dtype=f32, device="cuda")
    ops.attn_prefill_lse(q.view(B, nh, S, hd), kc, vt, B, S, nh, hd, att, lse)
    dbuf = (torch.randn(B * S, H + 64, generator=g, device="cuda") * 0.1).to(bf)      # token-major with a wider pitch
    datt = dbuf[:, :H]
    half = hd // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, half, device="cuda", dtype=f32) / half))
    ang = torch.arange(0, max(512, smax), device="cuda", dtype=f32)[:, None] * inv[None, :]
    cos_t, sin_t = ang.cos().contiguous(), ang.sin().contiguous()
    # separate passes
    dO = torch.empty(B * nh, S, hd, dtype=bf, device="cuda")
    ops.split_heads16(datt.contiguous(), B, S, nh, hd, dO)
    dq, dk, dv = (torch.empty(B * nh, S, hd, dtype=f32, device="cuda") for _ in range(3))
    dsum = torch.empty(B * nh, S, dtype=f32, device="cuda")
    ops.attn_backward(q, kc, v_rm, dO, att, lse, dsum, B, S, nh, hd, dq, dk, dv)
    ref = torch.empty(B * S, 3 * H, dtype=bf, device="cuda")
    ops.rope_merge_bwd(dq, dk, dv, cos_t, sin_t, B, S, nh, hd, 0, ref)
    # fused
    got = torch.full((B * S, 3 * H), 7.0, dtype=bf, device="cuda")
    dsum2 = torch.empty_like(dsum)
    ops.attn_backward_fused(q, kc, v_rm, datt, att, lse, dsum2, B, S, nh, hd, cos_t, sin_t, 0, got)
    assert torch.equal(dsum, dsum2)
    assert torch.equal(got, ref), (got.float() - ref.float()).abs().max().item()
    assert ref.float().abs().max().item() > 0
