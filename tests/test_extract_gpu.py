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


# ---------------------------------------------------------------------------------------------
# File-level entry of the audio boundary: wav files in, ``.npy`` out, upstream checkpoint
# files in -- jukebox/main.py:29-45,133-173,176-254 and jukebox/dataflow_inference.py:73-158.
# ---------------------------------------------------------------------------------------------
def _write_wav(path, audio, sr=44100, dtype=np.int16, channels=1):
    from scipy.io import wavfile

    x = audio / max(1e-9, np.abs(audio).max()) * 0.8
    if channels == 2:
        x = np.stack([x, x * 0.5], axis=1)
    if dtype == np.int16:
        x = (x * 32767).astype(np.int16)
    wavfile.write(path, sr, x.astype(dtype))


def test_get_acts_from_file_equals_in_memory_path(model, tmp_path):
    from llark_amd.jukebox import extract as E
    w, hps, vqvae, top_prior = model
    p = tmp_path / "clip.wav"
    _write_wav(p, synthetic_clip(3, seconds=1.2), channels=2)
    audio = E.load_audio_from_file(str(p))
    assert audio.ndim == 1 and np.abs(audio).max() == 1.0
    want = E.get_acts_from_audio(audio.copy(), hps, vqvae, top_prior, meanpool=True, pool_frames_per_second=10)
    got = E.get_acts_from_file(str(p), hps, vqvae, top_prior, meanpool=True, pool_frames_per_second=10)
    assert np.array_equal(got, want)
    assert got.shape[1] == hps.prior_width and got.shape[0] >= 1


def test_load_model_from_upstream_checkpoint_files(model, tmp_path):
    """``load_model()`` reads ``vqvae.pth.tar`` / ``prior_level_2.pth.tar`` written with upstream key names, DDP prefix and
    TWICE the layers the model is built with (5b: 72 on disk, prior_depth = 36): identical embeddings to the in-memory path."""
    from llark_amd.jukebox import extract as E
    import test_jukebox_checkpoint_cpu as TC
    w, hps, vqvae, top_prior = model
    depth = hps.prior_depth
    vq_ck, pr_ck, w2 = TC.upstream_style_checkpoints(hps, ckpt_depth=2 * depth, base=w)       # the fixture's weights + 'depth' more layers
    root = tmp_path / "jukebox" / "models"
    (root / "5b").mkdir(parents=True)
    torch.save(vq_ck, root / "5b" / "vqvae.pth.tar")
    torch.save(pr_ck, root / "5b" / "prior_level_2.pth.tar")
    import os
    old = os.environ.get("JUKEBOX_CACHE")
    os.environ["JUKEBOX_CACHE"] = str(tmp_path)
    try:
        hps2, vq2, tp2 = E.load_model("5b", hps=hparams_tiny(), device="cuda")          # weights=None: the checkpoint door
    finally:
        if old is None:
            del os.environ["JUKEBOX_CACHE"]
        else:
            os.environ["JUKEBOX_CACHE"] = old
    assert tp2.prior.depth == depth
    audio = synthetic_clip(4, seconds=1.6)
    audio = audio / np.abs(audio).max()
    a = E.get_acts_from_audio(audio.copy(), hps, vqvae, top_prior, True, 10)
    b = E.get_acts_from_audio(audio.copy(), hps2, vq2, tp2, True, 10)
    assert np.array_equal(a, b), "checkpoint-file path and in-memory weights disagree"
    with pytest.raises(FileNotFoundError, match="prior_level_2.pth.tar"):
        E.load_model("5b", hps=hparams_tiny(), device="cuda", restore_prior=str(tmp_path / "nope.pth.tar"))


def test_model_wrapper_handler_and_cli(model, tmp_path, monkeypatch):
    from llark_amd.jukebox import dataflow_inference as D
    from llark_amd.jukebox import extract as E
    w, hps, vqvae, top_prior = model
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    _write_wav(indir / "a.wav", synthetic_clip(5, seconds=1.6))
    _write_wav(indir / "b.wav", synthetic_clip(6, seconds=0.8), sr=22050)            # resampled + padded clip
    (indir / "empty.wav").write_bytes(b"")
    wrapper = D.JukeboxModelWrapper("5b", "cuda", weights=w, hps=hparams_tiny())
    assert wrapper.chunk_size == 32 and wrapper.max_batch_size == 16
    rep = wrapper(str(indir / "a.wav"))
    want = E.get_acts_from_file(str(indir / "a.wav"), hps, vqvae, top_prior, meanpool=True, pool_frames_per_second=10)
    assert np.array_equal(rep, want)
    assert wrapper(str(indir / "empty.wav")) is None                                  # EmptyFileError -> None (:113-115)
    handler = D.JukeboxModelHandler("5b", weights=w, hps=hparams_tiny())
    res = handler.run_inference([str(indir / "a.wav"), str(indir / "empty.wav")], wrapper)
    assert res[0].example.endswith("a.wav") and np.array_equal(res[0].inference[0], want) and res[1].inference == [None]
    n = D.run_files(str(indir), str(outdir), handler)
    assert n == 2 and sorted(p.name for p in outdir.iterdir()) == ["a.npy", "b.npy"]
    assert np.array_equal(np.load(outdir / "a.npy")[0], want)
    # the CLI of jukebox/main.py:203-254 (one .npy per wav; an empty file is an error there, as in the reference)
    (indir / "empty.wav").unlink()
    monkeypatch.setattr(E, "load_model", lambda *a, **k: (hps, vqvae, top_prior))
    out2 = tmp_path / "out2"
    E.main(["--input_dir", str(indir), "--output_dir", str(out2), "--pool-frames-per-second", "10", "--batch_size", "1", "--batch_idx", "0"])
    assert [p.name for p in out2.iterdir()] == ["a.npy"]
    assert np.array_equal(np.load(out2 / "a.npy"), want)
    E.main(["--input_dir", str(indir), "--output_dir", str(out2), "--pool-frames-per-second", "0"])
    assert np.load(out2 / "b.npy").shape == (hps.prior_width,)
