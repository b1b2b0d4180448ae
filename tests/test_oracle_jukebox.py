"""CPU: pins the oracle itself.  The reference ships no golden vectors for the Jukebox half
(PARITY UNPINNED, see oracle/jukebox_ref.py), so the oracle is checked (a) C restatement vs torch
restatement, (b) against the constants / rules the reference does pin, and
(c) through structural properties of the upstream algorithm."""
import math

import numpy as np
import pytest
import torch

from llark_amd.jukebox.hparams import hparams_5b, hparams_tiny
from llark_amd.jukebox.synthetic import init_codebook_from_encodings, make_jukebox_weights, synthetic_clip
from oracle import jukebox_c as C
from oracle import jukebox_ref as R


@pytest.fixture(scope="module")
def tiny():
    hps = hparams_tiny()
    w = make_jukebox_weights(hps, 0)
    calib = np.concatenate([C.encoder_forward(w, _clip(hps, 100 + i)[None], hps) for i in range(2)], axis=1)
    w["bottleneck.level_blocks.2.k"] = init_codebook_from_encodings(torch.from_numpy(calib), hps.l_bins)
    return hps, w


def _clip(hps, i, seconds=1.6):
    a = R.normalize_audio(synthetic_clip(i, seconds=seconds))[: hps.sample_length]
    return np.pad(a, (0, max(0, hps.sample_length - len(a)))).astype(np.float32)


def test_reference_constants():
    # jukebox/main.py:10-22
    assert R.T == 8192 and R.JUKEBOX_EXPECTED_SAMPLES_LEN == 1048576
    assert abs(R.ACTS_SAMPLE_RATE - 344.53125) < 1e-6
    hps = hparams_5b()
    hps.check()
    assert hps.raw_to_tokens == 128 and hps.n_state == 1200 and hps.head_dim == 150 and hps.block_ctx == 64
    # jukebox/main.py:162: frame_len = floor(345.65/10) = 34 -> 240 frames (ceil_mode False)
    assert math.floor(R.ACTS_SAMPLE_RATE / 10) == 34 and 8192 // 34 == 240


def test_c_matches_torch_encoder(tiny):
    hps, w = tiny
    a = _clip(hps, 0)
    xe_t = R.vqvae_encoder_forward(w, torch.from_numpy(a)[None, None, :], hps)[0].numpy()
    xe_c = C.encoder_forward(w, a[None], hps)
    assert xe_c.shape == (hps.emb_width, hps.n_ctx)
    scale = np.abs(xe_t).max()
    assert np.abs(xe_c - xe_t).max() <= 2e-5 * scale
    codes_c, enc, mind = C.encode_codes(w, a[None], hps, return_all=True)
    codes_t = R.bottleneck_encode(w["bottleneck.level_blocks.2.k"], torch.from_numpy(enc)).numpy()
    assert (codes_c == codes_t).mean() >= 0.995        # identical up to fp32 near-ties
    assert len(np.unique(codes_c)) > 50
    z = R.get_z(a, w, hps)
    assert z.shape == (1, hps.n_ctx) and (z.numpy() == codes_c).mean() >= 0.97


def test_conv_edges_and_layers():
    # single layers with odd lengths, both paddings, all dilations
    g = torch.Generator().manual_seed(3)
    for (cin, cout, k, s, p, d, tin) in [(1, 32, 4, 2, 1, 1, 1000), (32, 32, 3, 1, 27, 27, 777), (32, 64, 3, 1, 1, 1, 129),
                                          (64, 32, 4, 2, 1, 1, 514), (32, 32, 3, 1, 9, 9, 64)]:
        x = torch.randn(cin, tin, generator=g)
        wt = torch.randn(cout, cin, k, generator=g) * 0.2
        b = torch.randn(cout, generator=g)
        ref = torch.nn.functional.conv1d(x[None], wt, b, stride=s, padding=p, dilation=d)[0].numpy()
        got = C.conv1d(x.numpy(), wt.numpy(), b.numpy(), stride=s, pad=p, dil=d)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_get_z_errors(tiny):
    hps, w = tiny
    with pytest.raises(AssertionError):
        R.get_z(np.zeros(hps.sample_length - 1, dtype=np.float32), w, hps)


def test_cond_shapes_and_bins(tiny):
    hps, w = tiny
    x_cond, y_cond = R.get_cond(w, hps)
    assert x_cond.shape == (1, hps.n_ctx, hps.prior_width) and y_cond.shape == (1, 1, hps.prior_width)
    # offset 0, length = n_ctx*raw_to_tokens: the relative-position bins are monotone from bin 0
    rel = w["y_emb.relative_pos_emb.emb.weight"]
    tot = w["y_emb.total_length_emb.emb.weight"]
    ab = w["y_emb.absolute_pos_emb.emb.weight"]
    assert torch.allclose(x_cond[0, 0], tot[0] + ab[0] + rel[0])


def test_prior_causality_and_prev_block(tiny):
    hps, w = tiny
    torch.manual_seed(0)
    z = torch.randint(0, hps.l_bins, (1, hps.n_ctx))
    x_cond, y_cond = R.get_cond(w, hps)
    a0 = R.get_final_activations(z, x_cond, y_cond, w, hps)
    # changing z[t] never changes acts[<= t] (autoregressive: token t enters at position t+1)
    t = 200
    z2 = z.clone()
    z2[0, t] = (z2[0, t] + 1) % hps.l_bins
    a1 = R.get_final_activations(z2, x_cond, y_cond, w, hps)
    assert torch.equal(a0[0, : t + 1], a1[0, : t + 1])
    assert not torch.equal(a0[0, t + 1], a1[0, t + 1])
    # prev_block_attn: block 0 attends to zero-padded K/V -> exactly zero
    q = torch.randn(1, hps.n_ctx, hps.n_state)
    out = R.factored_attention(q, q.clone(), q.clone(), 3, hps.heads, hps.block_ctx)
    assert torch.equal(out[0, : hps.block_ctx], torch.zeros(hps.block_ctx, hps.n_state))


def test_pooling_rules(tiny):
    hps, w = tiny
    acts = torch.randn(hps.n_ctx, 4800)
    out = R.windowed_average(acts, 34)
    assert out.shape == (1, hps.n_ctx // 34, 4800)
    assert torch.allclose(out[0, 1], acts[34:68].mean(0), atol=1e-6)
    # slice rule main.py:136,154 and both pooling modes through get_acts_from_audio
    audio = synthetic_clip(5, seconds=1.0)                     # shorter than sample_length -> padded
    latent = math.floor(hps.n_ctx * len(audio) / hps.sample_length)
    rate = hps.n_ctx / (hps.sample_length / hps.sr)
    fl = math.floor(rate / 10)
    e10 = R.get_acts_from_audio(audio, w, hps, meanpool=True, pool_frames_per_second=10, depth=1)
    e0 = R.get_acts_from_audio(audio, w, hps, meanpool=True, pool_frames_per_second=0, depth=1)
    assert e10.shape == (latent // fl, hps.prior_width) and e0.shape == (hps.prior_width,)
