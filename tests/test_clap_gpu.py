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


def _mel_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clap_mel.npz"))


def test_logmel_kernel_matches_oracle_and_feature_extractor():
    """Fused STFT + mel + log kernel (fp32 radix-2 FFT) vs the float64 oracle and the golden of transformers'
    ClapFeatureExtractor.  Tolerance in dB: 2e-3 everywhere (measured 2.4e-4, quiet bands included: the FFT's error grows
    with log N where the reference's fp32 DFT-matrix conv grows with sqrt N); silent frames clamp to -100 dB."""
    from llark_amd.clap import ClapFrontend
    z = _mel_golden()
    wave = CR.fit_length(z["wave"])
    fe = ClapFrontend("cuda:0")
    got = fe.logmel(torch.from_numpy(wave)[None].cuda(), quantize_int16=False)
    assert got.shape == (1, 1, 1001, 64)
    got = got[0, 0].cpu().numpy().astype(np.float64)
    ref = CR.logmel(wave)
    d = np.abs(got - ref)
    loud = ref >= ref.max() - 70.0
    print(f"logmel max|d| loud {d[loud].max():.2e} dB, quiet {d[~loud].max():.2e} dB, vs HF golden {np.abs(got - z['logmel']).max():.2e}")
    assert d[loud].max() <= 2e-3 and d[~loud].max() <= 2e-3
    assert np.abs(got - z["logmel"])[loud].max() <= 2e-3
    assert np.abs(got[ref == -100.0] + 100.0).max() <= 1e-5 and (ref == -100.0).sum() > 64 * 50
    # int16 round trip folded into the sample read == the host round trip
    gq = fe.logmel(torch.from_numpy(wave)[None].cuda(), quantize_int16=True)[0, 0].cpu().numpy()
    gh = fe.logmel(torch.from_numpy(CR.quantize_roundtrip(wave))[None].cuda(), quantize_int16=False)[0, 0].cpu().numpy()
    assert np.abs(gq - gh).max() <= 1e-4
    # batch rows are independent; a clip that is not a multiple of the hop
    w2 = np.stack([wave[:100000], wave[5000:105000]])
    g2 = fe.logmel(torch.from_numpy(w2).cuda(), quantize_int16=False)[:, 0].cpu().numpy()
    assert g2.shape == (2, 100000 // 480 + 1, 64)
    r1 = CR.logmel(w2[1])
    assert np.abs(g2[1] - r1)[r1 >= r1.max() - 70].max() <= 2e-3


def test_module_waveform_to_embedding_like_the_reference_handler(tmp_path):
    """HipClapModule stands where laion_clap.CLAP_Module stands in clap_embeddings.py: load_ckpt (laion names, fused qkv,
    'state_dict' wrapper) -> model.get_audio_embedding([{'waveform': ...}]) -> (B, 512-like) unit vectors; against the
    oracle run from the float64 log-mel.  Also the raw-clip convenience entry with ragged lengths."""
    from clap_util import to_laion_names
    from llark_amd.clap import ClapDims, HipClapModule, load_audio_input
    spec = CR.ClapSpec(**TINY)
    w = CR.make_weights(spec, seed=5)
    ck = tmp_path / "clap.pt"
    torch.save({"epoch": 15, "state_dict": to_laion_names(w)}, ck)
    m = HipClapModule(enable_fusion=False, amodel="HTSAT-base", device="cuda:0", dims=ClapDims(**TINY))
    with pytest.raises(RuntimeError, match="no weights"):
        m.model.get_audio_embedding([{"waveform": torch.zeros(480000)}])
    m.load_ckpt(str(ck))
    z = _mel_golden()
    rng = np.random.default_rng(1)
    clips = [z["wave"], (rng.standard_normal(30000) * 0.2).astype(np.float32)]
    elems = [load_audio_input({"waveform": c}) for c in clips]
    out = m.model.get_audio_embedding([e["audio_features"][0] for e in elems]).cpu()
    feats = np.stack([CR.logmel(CR.fit_length(CR.quantize_roundtrip(c))) for c in clips]).astype(np.float32)
    ref = CR.forward(w, spec, torch.from_numpy(feats)[:, None])
    assert out.shape == (2, 64) and _rel(out, ref) <= 1e-4, _rel(out, ref)
    # raw clips: one longer than 10 s (rand_trunc at the offset the seeded generator draws), one shorter (repeatpad)
    long = (rng.standard_normal(500000) * 0.1).astype(np.float32)
    got = m.get_audio_embedding_from_data([long, clips[1]], rng=np.random.default_rng(9))
    off = int(np.random.default_rng(9).integers(0, 500000 - 480000 + 1))
    f2 = np.stack([CR.logmel(CR.fit_length(CR.quantize_roundtrip(long), offset=off)), feats[1]]).astype(np.float32)
    assert isinstance(got, np.ndarray) and _rel(torch.from_numpy(got), CR.forward(w, spec, torch.from_numpy(f2)[:, None])) <= 1e-4
    with pytest.raises(ValueError, match="480000"):
        m.model.get_audio_embedding([{"waveform": torch.zeros(1000)}])
    with pytest.raises(NotImplementedError):
        HipClapModule(enable_fusion=True)


def test_config5_clap_into_mpt_matches_oracle_pipeline():
    """BASELINE configs[4] in small: waveform -> HIP log-mel -> HIP HTSAT -> (B, 1, P) embedding -> HIP MPT (projector splice,
    F = 1) logits, against oracle log-mel -> oracle HTSAT -> oracle MPT.  Tolerance 1e-3 relative on the logits (north_star)."""
    from llark_amd.clap import ClapDims, ClapFrontend, HipClapAudioEncoder
    from llark_amd.m2t.mpt_engine import HipMptEngine, MptDims
    from oracle import mpt_ref as MR
    cspec = CR.ClapSpec(**TINY)
    cw = CR.make_weights(cspec, seed=5)
    mspec = MR.MptSpec(d_model=256, n_heads=2, n_layers=2, expansion_ratio=4, vocab_size=96, max_seq_len=128, mm_hidden_size=64,
                       audio_start_token=93, audio_end_token=94, audio_patch_token=95)
    mw = MR.make_weights(mspec, seed=7)
    rng = np.random.default_rng(3)
    waves = np.stack([CR.fit_length((rng.standard_normal(n) * 0.2).astype(np.float32)) for n in (90000, 480000)])
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 90, (2, 21), generator=g)
    ids[:, 1], ids[:, 2], ids[:, 3] = 93, 95, 94
    # oracle pipeline
    feats = np.stack([CR.logmel(CR.quantize_roundtrip(w)) for w in waves]).astype(np.float32)
    aud_ref = CR.forward(cw, cspec, torch.from_numpy(feats)[:, None]).unsqueeze(1)
    ref = MR.forward(mw, mspec, ids, aud_ref)["logits"]
    # HIP pipeline (embeddings stay on the GPU)
    fe = ClapFrontend("cuda:0")
    enc = HipClapAudioEncoder(cw, ClapDims(**TINY), "cuda:0")
    aud = enc.embed(fe.logmel(torch.from_numpy(waves).cuda(), quantize_int16=True)).unsqueeze(1)
    dims = MptDims(d_model=256, n_heads=2, n_layers=2, expansion_ratio=4, vocab_size=96, max_seq_len=128, mm_hidden_size=64)
    eng = HipMptEngine(dims, "cuda", 2, 64, precision="split")
    eng.load_state_dict(mw)
    logits = eng.forward_tokens(ids.cuda(), [(b, 1, aud[b]) for b in range(2)]).cpu()      # (clip, index of the start token, frames)
    assert _rel(aud.cpu(), aud_ref) <= 1e-4
    assert _rel(logits, ref) <= 1e-3, _rel(logits, ref)


def test_embed_cli_writes_one_npy_per_wav(tmp_path):
    """python -m llark_amd.clap.embed_cli (clap_embeddings.py's job): HTSAT-base-shaped checkpoint under laion names ->
    <name>.npy of shape (1, 512), equal to the module called directly on the same clips."""
    from scipy.io import wavfile
    from clap_util import to_laion_names
    from llark_amd.clap import HipClapModule, random_state_dict
    from llark_amd.clap.embed_cli import main, read_wav_48k
    sd = random_state_dict(seed=4)
    torch.save({"state_dict": to_laion_names(sd)}, tmp_path / "ck.pt")
    rng = np.random.default_rng(0)
    (tmp_path / "in").mkdir()
    wavfile.write(tmp_path / "in" / "x.wav", 48000, (rng.standard_normal(60000) * 4000).astype(np.int16))
    wavfile.write(tmp_path / "in" / "y.wav", 48000, (rng.standard_normal(20000) * 0.1).astype(np.float32))
    assert main(["--input-dir", str(tmp_path / "in"), "--output-dir", str(tmp_path / "out"), "--ckpt-file", str(tmp_path / "ck.pt"), "--batch-size", "2"]) == 0
    m = HipClapModule(device="cuda:0")
    m.load_state_dict(sd)
    want = m.get_audio_embedding_from_data([read_wav_48k(str(tmp_path / "in" / n)) for n in ("x.wav", "y.wav")])
    for i, n in enumerate(("x", "y")):
        e = np.load(tmp_path / "out" / f"{n}.npy")
        assert e.shape == (1, 512) and e.dtype == np.float32 and abs(np.linalg.norm(e) - 1) < 1e-5
        assert np.abs(e[0] - want[i]).max() <= 1e-6


def test_recorded_launch_list_replay_is_bit_identical_and_survives_shape_changes():
    """embed() records its launches on the first call per (batch, frames) and replays them afterwards: replays equal the
    recorded pass bit for bit, follow the input (no stale buffers), and a later, larger batch (workspaces regrow) neither
    breaks older shapes nor reuses their stale lists."""
    spec, w, eng = _engine(TINY, 5)
    g = torch.Generator().manual_seed(8)
    xa = (torch.randn(2, 1, 1001, 64, generator=g) * 20 - 30).cuda()
    xb = (torch.randn(2, 1, 1001, 64, generator=g) * 20 - 30).cuda()
    first = eng.embed(xa).cpu()                                   # records
    assert len(eng._plans) == 1
    assert torch.equal(eng.embed(xa).cpu(), first)                # replays
    rb = eng.embed(xb).cpu()
    assert not torch.equal(rb, first) and _rel(rb, CR.forward(w, spec, xb.cpu())) <= 1e-4
    x5 = (torch.randn(5, 1, 700, 64, generator=g) * 20 - 30).cuda()
    r5 = eng.embed(x5).cpu()                                      # bigger batch: workspaces regrow, version bumps
    assert _rel(r5, CR.forward(w, spec, x5.cpu())) <= 1e-4
    assert torch.equal(eng.embed(xa).cpu(), first)                # old shape re-records against the new buffers
    assert torch.equal(eng.embed(xa).cpu(), first) and torch.equal(eng.embed(x5).cpu(), r5)
    eng.replay = False
    assert torch.equal(eng.embed(xa).cpu(), first)                # the plain layer loop gives the same bits


def test_gemm_act_epilogue_and_duplicate_plane():
    """llark_gemm16_act: acc + bias -> exact GELU -> bf16 hi / lo (+ second hi copy written into a wider buffer) against
    torch fp32, both the split (hi + lo) and the single-plane (OUT16) forms; M large enough for the persistent tile path too."""
    from llark_amd import ops as O
    g = torch.Generator().manual_seed(1)
    for m, n, k in ((300, 96, 64), (20000, 256, 128)):
        a = torch.randn(m, k, generator=g).cuda()
        w = (torch.randn(n, k, generator=g) * 0.2).cuda()
        bias = torch.randn(n, generator=g).cuda()
        a_hi, a_lo = O.split16(a, torch.bfloat16)
        wt = O.pack_weight16(w, False, torch.bfloat16)
        ref = torch.nn.functional.gelu((a_hi.float() + a_lo.float()).double() @ wt.float().double().t() + bias.double())
        out3 = torch.zeros(m, 3 * n, dtype=torch.bfloat16, device="cuda")
        O.gemm16_act(a_hi, a_lo, wt, bias, n, out3[:, :n], out3[:, n:2 * n], out3[:, 2 * n:], act=2)
        got = out3[:, :n].float() + out3[:, n:2 * n].float()
        assert torch.equal(out3[:, 2 * n:], out3[:, :n])
        assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
        one = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        O.gemm16_act(a_hi, None, wt, bias, n, one, act=2)
        ref1 = torch.nn.functional.gelu(a_hi.float().double() @ wt.float().double().t() + bias.double())
        assert (one.double() - ref1).abs().max().item() <= 6e-3 * ref1.abs().max().item()
        plain = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        O.gemm16_act(a_hi, None, wt, bias, n, plain, act=0)
        assert (plain.double() - (a_hi.float().double() @ wt.float().double().t() + bias.double())).abs().max().item() <= 6e-3 * ref1.abs().max().item() + 0.05


@pytest.mark.parametrize("kcat,fuse", [("0", "1"), ("1", "0"), ("1", "1")])
def test_linear_forms_agree_with_oracle(monkeypatch, kcat, fuse):
    """The two-launch (split + W_lo correction) and one-launch (K-concatenated) linears, with and without the fused GELU
    epilogue, all meet the fp32-class bar."""
    monkeypatch.setenv("LLARK_CLAP_KCAT", kcat)
    monkeypatch.setenv("LLARK_CLAP_FUSE_GELU", fuse)
    spec, w, eng = _engine(TINY, 5)
    assert eng.kcat == (kcat == "1")
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 1, 1001, 64, generator=g) * 20 - 30
    assert _rel(eng.embed(x.cuda()).cpu(), CR.forward(w, spec, x)) <= 1e-4
