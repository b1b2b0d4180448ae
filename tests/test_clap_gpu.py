"""GPU parity of the HIP CLAP HTSAT audio encoder (llark_amd/clap, csrc/clap.hip) against oracle/clap_ref.py and the golden
vectors of the independent transformers port (tests/golden/clap_tiny.npz).  Tolerances: fp32-class mode 1e-4 relative on the
embedding (north_star's embedding bar), bf16 mode 3e-2."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import clap_ref as CR

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clap_tiny.npz")
TINY = dict(embed_dim=32, depths=[2, 2, 2, 1], heads=[1, 2, 4, 8], proj_dim=64)


def _engine(spec_kw, seed, precision="fp32"):
    from llark_amd.clap import ClapDims, HipClapAudioEncoder
    spec = CR.ClapSpec(**spec_kw)
    w = CR.make_weights(spec, seed=seed)
    return spec, w, HipClapAudioEncoder(w, ClapDims(**spec_kw), device="cuda:0", precision=precision)


def _rel(a, b):
    return (a - b).abs().max().item() / b.abs().max().item()


def test_window_attention_kernel_matches_oracle_block():
    """clap_window_attn alone (shifted and unshifted, several map sizes) against the attention half of oracle swin_block."""
    from llark_amd import ops as O
    g = torch.Generator().manual_seed(3)
    for (B, H, W, C, heads, shift) in ((2, 16, 16, 64, 2, 0), (2, 16, 16, 64, 2, 4), (1, 32, 16, 128, 4, 4), (3, 8, 8, 32, 1, 0), (1, 64, 64, 128, 4, 4)):
        hd = C // heads
        qkv = torch.randn(B * H * W, 3 * C, generator=g)
        table = torch.randn(225, heads, generator=g)
        # oracle: same math as swin_block between the q/k/v linears and the output dense
        x = qkv.view(B, H, W, 3 * C)
        if shift:
            x = torch.roll(x, (-shift, -shift), (1, 2))
        win = CR._window_partition(x, 8)
        q, k, v = [win[..., i * C:(i + 1) * C].reshape(-1, 64, heads, hd).transpose(1, 2) for i in range(3)]
        att = q @ k.transpose(-1, -2) / math.sqrt(hd) + table[CR.relative_position_index(8).view(-1)].view(64, 64, heads).permute(2, 0, 1)
        if shift:
            hr = (torch.arange(H) >= H - 8).long() + (torch.arange(H) >= H - shift).long()
            wr = (torch.arange(W) >= W - 8).long() + (torch.arange(W) >= W - shift).long()
            mw = CR._window_partition((hr[None, :, None, None] * 3 + wr[None, None, :, None]).float(), 8).view(-1, 64)
            mask = mw.unsqueeze(1) - mw.unsqueeze(2)
            mask = mask.masked_fill(mask != 0, -100.0)
            nW = mask.shape[0]
            att = (att.view(B, nW, heads, 64, 64) + mask.view(1, nW, 1, 64, 64)).view(-1, heads, 64, 64)
        ctx = (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(-1, 64, C)
        ref = CR._window_reverse(ctx, 8, H, W)
        if shift:
            ref = torch.roll(ref, (shift, shift), (1, 2))
        ref = ref.reshape(B * H * W, C)
        hi = torch.empty(B * H * W, C, dtype=torch.bfloat16, device="cuda:0")
        lo = torch.empty_like(hi)
        O.clap_window_attn(qkv.cuda(), B, H, W, C, heads, 8, shift, table.cuda(), hi, lo)
        got = (hi.float() + lo.float()).cpu()
        assert (got - ref).abs().max().item() <= 3e-5 * ref.abs().max().item(), (B, H, W, C, heads, shift)


def test_tiny_matches_oracle_and_independent_port():
    spec, w, eng = _engine(TINY, 5)
    z = np.load(GOLD)
    x = torch.from_numpy(z["x"])
    got = eng.embed(x.cuda(), normalize=False).cpu()
    assert _rel(got, torch.from_numpy(z["audio_embeds"])) <= 1e-4                       # transformers.ClapAudioModelWithProjection
    assert _rel(got, CR.forward(w, spec, x, normalize=False)) <= 1e-4
    gn = eng.embed(x.cuda()).cpu()
    assert torch.allclose(gn.norm(dim=-1), torch.ones(2), atol=1e-5)
    assert _rel(gn, CR.forward(w, spec, x)) <= 1e-4
    assert torch.equal(eng.embed(x[:, 0].cuda()).cpu(), gn)                             # (B, frames, mel) accepted; deterministic


def test_htsat_base_widths_match_oracle():
    """The reference's HTSAT-base shape (embed 128, depths 2/2/12/2, heads 4/8/16/32, 512-d projection), seeded weights."""
    spec, w, eng = _engine({}, 11)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 1, 1001, 64, generator=g) * 20 - 30
    ref = CR.forward(w, spec, x)
    got = eng.embed(x.cuda()).cpu()
    assert got.shape == (2, 512)
    assert _rel(got, ref) <= 1e-4, _rel(got, ref)


def test_bf16_mode_and_ragged_frames():
    spec, w, eng = _engine(TINY, 7, precision="bf16")
    g = torch.Generator().manual_seed(4)
    for frames in (1024, 1001, 300):                                                    # exact fit, the 10 s clip, a short clip
        x = torch.randn(3, 1, frames, 64, generator=g) * 20 - 30
        ref = CR.forward(w, spec, x)
        got = eng.embed(x.cuda()).cpu()
        assert _rel(got, ref) <= 3e-2, (frames, _rel(got, ref))
    spec, w, eng32 = _engine(TINY, 7)
    x = torch.randn(1, 1, 300, 64, generator=g) * 20 - 30
    assert _rel(eng32.embed(x.cuda()).cpu(), CR.forward(w, spec, x)) <= 1e-4


def test_errors_are_loud():
    spec, w, eng = _engine(TINY, 5)
    with pytest.raises(ValueError, match="less than or equal to the swin input size"):
        eng.embed(torch.zeros(1, 1, 1025, 64, device="cuda:0"))
    with pytest.raises(ValueError, match="log-mel"):
        eng.embed(torch.zeros(1, 1, 1001, 32, device="cuda:0"))
    from llark_amd._lib import LlarkHipError
    with pytest.raises(LlarkHipError):
        eng.embed(torch.zeros(1, 1, 1001, 64))                                          # CPU tensor: there is no CPU path
