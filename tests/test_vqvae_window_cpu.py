"""CPU check of what the near-tie fix-up of the fused VQ-VAE encoder rests on (round 4): the exact encoder evaluated on a
window of `win_tokens` tokens around a token -- clamped to the clip edges -- gives, at that token, the SAME BITS as the exact
encoder on the whole clip.  Evaluated with the C oracle (oracle/jukebox_ref.c = the arithmetic the exact HIP kernels are
bit-equal to), so the receptive-field arithmetic of llark_amd.jukebox.vqvae.receptive_halo_tokens and the window placement of
csrc/vqvae.hip gather_windows_kernel are pinned without a GPU."""
import numpy as np
import torch

from llark_amd.jukebox.hparams import hparams_5b, hparams_tiny
from llark_amd.jukebox.synthetic import make_vqvae_weights, synthetic_clip
from llark_amd.jukebox.vqvae import receptive_halo_tokens


def _window_start(tok, halo, win, n_ctx):
    return min(max(tok - halo, 0), n_ctx - win)


def test_receptive_halo_matches_hand_count():
    # 5b: left = right = 10455 samples (DESIGN section 4 "VQ-VAE": 1 + 40 per resnet, x2 (+1 / +2) per strided conv)
    assert receptive_halo_tokens(hparams_5b()) == 82
    assert receptive_halo_tokens(hparams_tiny()) == 82


def test_window_value_is_bit_equal_to_clip_value():
    from oracle import jukebox_c as C
    from oracle import jukebox_ref as R

    hps = hparams_tiny()
    w = make_vqvae_weights(hps, 0)
    a = R.normalize_audio(synthetic_clip(3, seconds=1.6))[: hps.sample_length]
    a = np.pad(a, (0, hps.sample_length - len(a))).astype(np.float32)
    full = C.encoder_forward(w, a[None], hps)                         # one clip in, (64, n_ctx) out
    halo = receptive_halo_tokens(hps)
    win = -(-(2 * (halo + 2) + 1) // 8) * 8
    r2t = hps.raw_to_tokens
    assert win < hps.n_ctx
    for tok in (0, 1, halo - 1, halo, halo + 1, 200, 257, hps.n_ctx - halo - 1, hps.n_ctx - 2, hps.n_ctx - 1):
        s = _window_start(tok, halo + 2, win, hps.n_ctx)
        xw = C.encoder_forward(w, a[None, s * r2t:(s + win) * r2t], hps)
        assert xw.shape == (hps.emb_width, win)
        assert np.array_equal(xw[:, tok - s], full[:, tok]), f"token {tok} (window start {s}): window value differs from the clip value"
    # one token LESS of halo than the receptive field needs must show (the bound is tight, not padded by luck)
    tok = 250
    s = tok - (halo - 2)
    xw = C.encoder_forward(w, a[None, s * r2t:(s + win) * r2t], hps)
    assert not np.array_equal(xw[:, tok - s], full[:, tok])
