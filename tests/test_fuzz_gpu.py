"""Seeded randomised parity sweeps (shapes drawn at random, fixed seeds => reproducible): the GEMM with every epilogue /
dtype / main loop, the attention kernels with and without ALiBi and lo planes, the row-norm kernels, and the VQ-VAE
conv / residual kernels (bit-exact vs the C oracle).  These complement the fixed-shape tests with ragged sizes around every
tile boundary (128 / 256 rows and columns, 64-key attention tiles, 256-sample conv tiles)."""
import random

import numpy as np
import pytest
import torch

from conftest import report_close

pytestmark = pytest.mark.gpu


def _cases(seed, n, draw):
    rng = random.Random(seed)
    return [draw(rng) for _ in range(n)]


GEMM_CASES = _cases(1, 28, lambda r: dict(m=r.choice([1, 7, 16, 17, 100, 127, 128, 129, 255, 256, 257, 500, 700, 1100]),
                                           n=r.choice([32, 64, 96, 128, 192, 256, 320, 448, 512, 576]),
                                           k=64 * r.randint(1, 9), bf=r.random() < 0.5, split=r.random() < 0.5,
                                           epi=r.choice(["f32", "resid", "qgelu", "split16", "out16", "swiglu16", "swiglu_split"]),
                                           frag=r.random() < 0.4))


@pytest.mark.parametrize("c", GEMM_CASES, ids=lambda c: "m{m}n{n}k{k}{e}{d}{s}{f}".format(e=c["epi"], d="bf" if c["bf"] else "f16",
                                                                                          s="S" if c["split"] else "", f="F" if c["frag"] else "", **c))
def test_gemm_random_shapes(c):
    from llark_amd import ops
    m, n, k = c["m"], c["n"], c["k"]
    dt = torch.bfloat16 if c["bf"] else torch.float16
    g = torch.Generator().manual_seed(m * 131 + n * 7 + k)
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(n, k, generator=g) * 0.1).to(dt)
    bias = torch.randn(n, generator=g)
    hi, lo = ops.split16(a.cuda(), dt, kmult=64)
    a_eff = (hi.float() + (lo.float() if c["split"] else 0.0)).cpu()[:, :k].double()
    wt = ops.pack_weight16(w.cuda(), False, dt, kmult=64)
    if c["frag"]:
        ops.attach_frag(wt, n)
    ref = a_eff @ w.double().t()
    bound = (a_eff.abs() @ w.double().abs().t()).max().item()
    tol = 4e-6 * bound + 1e-6
    l = lo if c["split"] else None
    epi = c["epi"]
    if epi in ("swiglu16", "swiglu_split"):
        if n % 64:
            pytest.skip("SwiGLU needs n % 64 == 0")
        inter = n // 2
        gate, up = ref.view(m, n // 64, 2, 32)[:, :, 0].reshape(m, inter), ref.view(m, n // 64, 2, 32)[:, :, 1].reshape(m, inter)
        want = torch.nn.functional.silu(gate) * up
        oh = torch.zeros((m, inter), dtype=dt, device="cuda")
        ol = torch.zeros_like(oh)
        ops.gemm16(hi, l, wt, None, n, ops.EPI_SWIGLU_SPLIT if epi == "swiglu_split" else ops.EPI_SWIGLU16, out_hi=oh,
                   out_lo=ol if epi == "swiglu_split" else None)
        got = oh.float() + (ol.float() if epi == "swiglu_split" else 0.0)
        rel = 2 ** -8 if epi == "swiglu16" else 2 ** -15
        report_close(epi, got.cpu(), want, rel * want.abs().max().item() + 8 * tol, rel)
        return
    want = ref + bias.double()
    if epi == "f32":
        out = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16(hi, l, wt, bias.cuda(), n, ops.EPI_F32, c=out)
        report_close("f32", out.cpu(), want, tol)
    elif epi == "resid":
        r0 = torch.randn(m, n, generator=g)
        out = r0.clone().cuda()
        ops.gemm16(hi, l, wt, bias.cuda(), n, ops.EPI_RESID, c=out, resid=out)
        report_close("resid", out.cpu(), want + r0.double(), tol + 1e-6)
    else:
        oh = torch.zeros((m, n), dtype=dt, device="cuda")
        ol = torch.zeros_like(oh)
        if epi == "qgelu":
            want = want * torch.sigmoid(1.702 * want)
            ops.gemm16(hi, l, wt, bias.cuda(), n, ops.EPI_QGELU_SPLIT, out_hi=oh, out_lo=ol)
            got = oh.float() + ol.float()
            prec = 2 ** -15
        elif epi == "split16":
            ops.gemm16(hi, l, wt, bias.cuda(), n, ops.EPI_SPLIT16, out_hi=oh, out_lo=ol)
            got = oh.float() + ol.float()
            prec = 2 ** -15
        else:
            ops.gemm16(hi, l, wt, bias.cuda(), n, ops.EPI_OUT16, out_hi=oh)
            got = oh.float()
            prec = 2 ** -8 if c["bf"] else 2 ** -10
        report_close(epi, got.cpu(), want, prec * want.abs().max().item() + 8 * tol, prec)


ATT_CASES = _cases(2, 16, lambda r: dict(B=r.randint(1, 3), nh=r.randint(1, 3), S=r.choice([1, 1, 2, 15, 63, 64, 65, 130, 200]),
                                          past=r.choice([0, 0, 1, 37, 64, 100]), split=r.random() < 0.5, alibi=r.random() < 0.5))


@pytest.mark.parametrize("c", ATT_CASES, ids=lambda c: "B{B}h{nh}S{S}p{past}{s}{a}".format(s="S" if c["split"] else "", a="A" if c["alibi"] else "", **c))
def test_attention_random_shapes(c):
    """prefill (S > 1) and decode (S = 1) kernels vs fp32 attention on the bf16-rounded operands, with / without ALiBi."""
    from llark_amd import ops
    B, nh, S, past, sp = c["B"], c["nh"], c["S"], c["past"], c["split"]
    hd, T = 128, c["past"] + c["S"]
    smax = ops.round_up(T + 5, 8)
    g = torch.Generator().manual_seed(B * 1000 + nh * 100 + S * 3 + past)
    q, k, v = (torch.randn(B, nh, n_, hd, generator=g) for n_ in (S, T, T))

    def planes(x):
        h = x.bfloat16()
        return h, (x - h.float()).bfloat16()

    qh, ql = planes(q)
    kh, kl = planes(k)
    vh, vl = planes(v.transpose(2, 3).contiguous())
    bf = dict(dtype=torch.bfloat16, device="cuda")
    kc, kcl, vc, vcl = (torch.zeros(shape, **bf) for shape in ((B, nh, smax, hd), (B, nh, smax, hd), (B, nh, hd, smax), (B, nh, hd, smax)))
    kc[:, :, :T], kcl[:, :, :T], vc[:, :, :, :T], vcl[:, :, :, :T] = kh.cuda(), kl.cuda(), vh.cuda(), vl.cuda()
    q_e = qh.float() + (ql.float() if sp else 0)
    k_e = kh.float() + (kl.float() if sp else 0)
    v_e = (vh.float() + (vl.float() if sp else 0)).transpose(2, 3)
    att = torch.matmul(q_e, k_e.transpose(2, 3)) * hd ** -0.5
    slopes = None
    if c["alibi"]:
        slopes = 1.0 / torch.pow(2, torch.arange(1, nh + 1, dtype=torch.float32) * (8 / nh))
        att = att + (torch.arange(T, dtype=torch.float32) - (T - 1)).view(1, 1, 1, T) * slopes.view(1, nh, 1, 1)
    mask = torch.full((S, T), float("-inf")).triu(diagonal=past + 1)
    ref = torch.matmul(torch.softmax(att + mask, dim=-1), v_e).transpose(1, 2).reshape(B * S, nh * hd)
    out, outl = torch.empty((B * S, nh * hd), **bf), torch.empty((B * S, nh * hd), **bf)
    args = (qh.cuda().contiguous(), kc, vc, B) + ((nh, hd, past + 1, out) if S == 1 else (S, nh, hd, past, out))
    kw = dict(q_lo=ql.cuda().contiguous(), k_cache_lo=kcl, vt_cache_lo=vcl, out_lo=outl) if sp else {}
    (ops.attn_decode if S == 1 else ops.attn_prefill)(*args, alibi_slopes=None if slopes is None else slopes.cuda(), **kw)
    got = out.float() + (outl.float() if sp else 0)
    tol = 3e-5 if sp else 1.2e-2                                     # single pass: P and O are rounded to bf16
    report_close("attention", got.cpu(), ref, tol * max(1.0, ref.abs().max().item()), tol)


def test_norm_kernels_random_widths():
    from llark_amd import ops
    rng = random.Random(3)
    for _ in range(12):
        rows, width = rng.randint(1, 40), 4 * rng.randint(8, 1200)
        g = torch.Generator().manual_seed(rows * 7 + width)
        x = torch.randn(rows, width, generator=g) * 3 + 0.5
        gam, bet = torch.randn(width, generator=g), torch.randn(width, generator=g)
        bf = dict(dtype=torch.bfloat16, device="cuda")
        hi, lo = torch.empty((rows, width), **bf), torch.empty((rows, width), **bf)
        ops.rmsnorm_bf16(x.cuda(), gam.cuda(), 1e-5, hi, lo)
        ref = gam * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))
        report_close(f"rmsnorm {rows}x{width}", (hi.float() + lo.float()).cpu(), ref, 2e-5 * ref.abs().max().item())
        ops.layernorm_bf16(x.cuda(), gam.cuda(), bet.cuda(), 1e-5, hi, lo)
        ref = torch.nn.functional.layer_norm(x, (width,), gam, bet, 1e-5)
        report_close(f"layernorm {rows}x{width}", (hi.float() + lo.float()).cpu(), ref, 2e-5 * ref.abs().max().item())
        if width <= 4800:
            h16, l16 = torch.empty((rows, width), dtype=torch.float16, device="cuda"), torch.empty((rows, width), dtype=torch.float16, device="cuda")
            ops.layernorm_split(x.cuda(), gam.cuda(), bet.cuda(), 1e-5, h16, l16)
            report_close(f"layernorm_split {rows}x{width}", (h16.float() + l16.float()).cpu(), ref, 2e-6 * ref.abs().max().item() + 1e-6)


def test_conv_kernels_random_lengths_bit_exact():
    """conv (k=4 s=2, k=3 s=1) and residual blocks at random lengths / dilations: bit-identical to the C oracle's fma order."""
    from llark_amd import ops
    from oracle import jukebox_c as C
    rng = random.Random(4)
    for _ in range(10):
        n, t, dil = rng.randint(1, 3), rng.choice([1, 2, 31, 255, 256, 257, 600, 1025, 3000]), rng.choice([1, 3, 9, 27])
        g = torch.Generator().manual_seed(t * 5 + dil)
        x = torch.randn(n, 32, t, generator=g)
        w1, b1 = torch.randn(32, 32, 3, generator=g) * 0.2, torch.randn(32, generator=g) * 0.1
        w2, b2 = torch.randn(32, 32, 1, generator=g) * 0.2, torch.randn(32, generator=g) * 0.1
        y = ops.resblock(x.cuda(), ops.pack_conv_weight(w1.cuda()), b1.cuda(), ops.pack_conv_weight(w2.cuda()), b2.cuda(), dil)
        npc = lambda a: np.ascontiguousarray(a.numpy())
        ref = torch.from_numpy(np.stack([C.resblock(npc(x[i]), npc(w1), npc(b1), npc(w2), npc(b2), dil) for i in range(n)]))
        assert torch.equal(y.cpu(), ref), f"resblock n={n} t={t} dil={dil}: max diff {(y.cpu() - ref).abs().max().item():.3e}"
        if t >= 4:
            wd, bd = torch.randn(32, 32, 4, generator=g) * 0.2, torch.randn(32, generator=g) * 0.1
            yd = ops.conv1d(x.cuda(), ops.pack_conv_weight(wd.cuda()), bd.cuda(), 2, 1)
            refd = torch.from_numpy(np.stack([C.conv1d(npc(x[i]), npc(wd), npc(bd), 2, 1, 1) for i in range(n)]))
            assert torch.equal(yd.cpu(), refd), f"down conv n={n} t={t}"
