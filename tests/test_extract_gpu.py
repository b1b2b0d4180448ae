"""GPU parity: the drop-in function set of jukebox/main.py end to end (audio -> pooled embedding)."""

import numpy as np
import pytest
import torch

from conftest import report_close
from llark_amd.jukebox.hparams import hparams_tiny
from llark_amd.jukebox.synthetic import init_codebook_from_encodings, make_jukebox_weights, synthetic_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from llark_amd.jukebox import extract as E
    from oracle import jukebox_c as C
    from oracle import jukebox_ref as R
    hps = hparams_tiny()
    w = make_jukebox_weights(hps, 0)
    a = R.normalize_audio(synthetic_clip(100, seconds=1.6))[: hps.sample_length]
    a = np.pad(a, (0, hps.sample_length - len(a))).astype(np.float32)
    w["bottleneck.level_blocks.2.k"] = init_codebook_from_encodings(
        torch.from_numpy(C.encoder_forward(w, a[None], hps)), hps.l_bins)
    return (w,) + E.load_model("5b", weights=w, hps=hps, device="cuda")


def test_get_acts_matches_oracle(model):
    from llark_amd.jukebox import extract as E
    from oracle import jukebox_ref as R
    w, hps, vqvae, top_prior = model
    for clip, seconds in [(0, 1.6), (1, 0.9)]:                      # full-length and padded (ragged) clip
        audio = R.normalize_audio(synthetic_clip(clip, seconds=seconds))
        for fps in (10, 0):
            ref = R.get_acts_from_audio(audio, w, hps, meanpool=True, pool_frames_per_second=fps)
            got = E.get_acts_from_audio(audio.copy(), hps, vqvae, top_prior, meanpool=True, pool_frames_per_second=fps)
            assert got.shape == ref.shape and got.dtype == np.float32
            report_close(f"embedding clip {clip} fps {fps}", got, ref, 1e-4 * np.abs(ref).max())


def test_batch_equals_single_and_module(model):
    from llark_amd.jukebox import extract as E
    from oracle import jukebox_ref as R
    w, hps, vqvae, top_prior = model
    audios = [R.normalize_audio(synthetic_clip(i, seconds=s)) for i, s in [(0, 1.6), (1, 0.9), (2, 1.5)]]
    batch = E.get_acts_from_audio_batch(audios, hps, vqvae, top_prior, True, 10)
    for a, b in zip(audios, batch):
        single = E.get_acts_from_audio(a.copy(), hps, vqvae, top_prior, True, 10)
        assert np.array_equal(single, b), "clips must not interact inside a batch"
    enc = E.WrappedAudioEncoder(hps=hps, weights=w, device="cuda")
    full = np.stack([np.pad(a, (0, max(0, hps.sample_length - len(a))))[: hps.sample_length] for a in audios]).astype(np.float32)
    out = enc(torch.from_numpy(full).cuda())
    fl = enc.frame_len
    assert out.shape == (3, hps.n_ctx // fl, hps.prior_width)
    assert np.array_equal(out[0].cpu().numpy(), batch[0])


def test_errors(model):
    from llark_amd.jukebox import extract as E
    w, hps, vqvae, top_prior = model
    with pytest.raises(AssertionError):
        E.get_z(np.zeros(hps.sample_length - 5, dtype=np.float32), vqvae)
    with pytest.raises(AssertionError):
        E.windowed_average(torch.zeros(4, 4, 4, device="cuda"), 2)
