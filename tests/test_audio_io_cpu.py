"""File-level entry of the audio boundary: ``load_audio_from_file`` (jukebox/main.py:29-45) on
wav files of every sample format, from paths and from the in-memory file objects the Beam worker passes
(jukebox/dataflow_inference.py:101-103), the empty-file error, and the shard selection of the CLI."""
import io
import os

import numpy as np
import pytest
from scipy.io import wavfile

from llark_amd.jukebox import extract as E


def _tone(n, sr, f=440.0, amp=0.4, seed=0):
    t = np.arange(n) / sr
    return (amp * np.sin(2 * np.pi * f * t) + 0.05 * np.random.default_rng(seed).standard_normal(n)).astype(np.float32)


def _expect(x_float_channels_last):
    """jukebox/main.py:36-45 on already-decoded float samples [(n,) or (n, channels)]."""
    a = x_float_channels_last.astype(np.float32)
    if a.ndim == 2:
        a = a.mean(axis=1)
    peak = np.abs(a).max()
    return (a / peak if peak > 0 else a).astype(np.float32)


def test_int16_mono_44100_is_bit_identical(tmp_path):
    x = (_tone(30000, 44100) * 32767).astype(np.int16)
    p = tmp_path / "a.wav"
    wavfile.write(p, 44100, x)
    got = E.load_audio_from_file(str(p))
    assert got.dtype == np.float32 and got.ndim == 1 and len(got) == len(x)
    np.testing.assert_array_equal(got, _expect(x.astype(np.float32) / 32768.0))
    assert np.abs(got).max() == 1.0


def test_float32_stereo_is_downmixed_then_normalised(tmp_path):
    x = np.stack([_tone(20000, 44100, 330.0, seed=1), _tone(20000, 44100, 550.0, 0.2, seed=2)], axis=1)
    p = tmp_path / "st.wav"
    wavfile.write(p, 44100, x)
    np.testing.assert_array_equal(E.load_audio_from_file(p), _expect(x))


@pytest.mark.parametrize("dtype,scale,offset", [(np.int32, 2.0 ** 31, 0.0), (np.uint8, 128.0, 128.0)])
def test_other_integer_formats_use_soundfile_scaling(tmp_path, dtype, scale, offset):
    f = _tone(5000, 44100)
    x = (f * (scale - 1) * 0.9 + offset).astype(dtype)
    p = tmp_path / "i.wav"
    wavfile.write(p, 44100, x)
    want = _expect((x.astype(np.float64) - offset).astype(np.float32) / np.float32(scale))
    np.testing.assert_allclose(E.load_audio_from_file(p), want, rtol=0, atol=1e-7)


def test_resampling_to_44100(tmp_path):
    sr = 22050
    x = (_tone(sr, sr, 440.0) * 32767).astype(np.int16)                  # one second
    p = tmp_path / "lo.wav"
    wavfile.write(p, sr, x)
    got = E.load_audio_from_file(p)
    assert len(got) == 44100 and np.abs(got).max() == 1.0
    # the 440 Hz tone survives: correlation with the ideal 44.1 kHz tone
    t = np.arange(44100) / 44100
    ref = np.sin(2 * np.pi * 440.0 * t)
    c = np.dot(got[2000:-2000], ref[2000:-2000]) / (np.linalg.norm(got[2000:-2000]) * np.linalg.norm(ref[2000:-2000]))
    assert c > 0.98
    x48 = (_tone(4800, 48000, 1000.0) * 32767).astype(np.int16)
    p48 = tmp_path / "hi.wav"
    wavfile.write(p48, 48000, x48)
    assert len(E.load_audio_from_file(p48)) == 4410


def test_file_object_input_like_the_beam_worker(tmp_path):
    x = (_tone(9000, 44100) * 20000).astype(np.int16)
    p = tmp_path / "b.wav"
    wavfile.write(p, 44100, x)
    from llark_amd.jukebox.dataflow_inference import read_wav_bytes

    got = E.load_audio_from_file(io.BytesIO(read_wav_bytes(str(p))))
    np.testing.assert_array_equal(got, E.load_audio_from_file(str(p)))
    with pytest.raises(NotImplementedError):
        read_wav_bytes("gs://bucket/a.wav")


def test_empty_and_silent_files(tmp_path):
    p = tmp_path / "empty.wav"
    p.write_bytes(b"")
    with pytest.raises(E.EmptyFileError, match="probably empty"):
        E.load_audio_from_file(str(p))
    with pytest.raises(E.EmptyFileError):
        E.load_audio_from_file(io.BytesIO(b""))
    z = tmp_path / "zero.wav"                                             # header only, zero samples
    wavfile.write(z, 44100, np.zeros(0, dtype=np.int16))
    with pytest.raises(E.EmptyFileError):
        E.load_audio_from_file(z)
    assert issubclass(E.EmptyFileError, ValueError)
    s = tmp_path / "silent.wav"                                           # all-zero samples: norm_factor == 0 branch (main.py:42)
    wavfile.write(s, 44100, np.zeros(100, dtype=np.int16))
    assert not E.load_audio_from_file(s).any()


def test_pad_and_shard_selection():
    a = np.ones(10, dtype=np.float32)
    assert len(E.maybe_pad_audio_to_max_len(a)) == E.JUKEBOX_EXPECTED_SAMPLES_LEN
    long = np.ones(E.JUKEBOX_EXPECTED_SAMPLES_LEN + 5, dtype=np.float32)
    assert E.maybe_pad_audio_to_max_len(long) is long                     # truncation happens in get_z (main.py:59)
    paths = [f"f{i}" for i in range(10)]
    assert E._select_shard(paths, None, None) == paths
    assert E._select_shard(paths, 4, 2) == ["f8", "f9"]
    with pytest.raises(ValueError, match="Invalid batch index"):
        E._select_shard(paths, 4, 3)


def test_file_list_and_result_writer(tmp_path):
    from llark_amd.jukebox import dataflow_inference as D

    for n in ("b.wav", "a.wav", "notes.txt"):
        (tmp_path / n).write_bytes(b"x")
    assert [os.path.basename(p) for p in D.get_input_file_list(str(tmp_path))] == ["a.wav", "b.wav"]
    out = tmp_path / "out"
    rep = np.arange(6, dtype=np.float32).reshape(2, 3)
    D.write_prediction_result(D.PredictionResult(str(tmp_path / "a.wav"), [rep]), str(out))
    np.testing.assert_array_equal(np.load(out / "a.npy"), rep[None])      # run_inference wraps the output in a list (:156)
    D.write_prediction_result(D.PredictionResult(str(tmp_path / "b.wav"), [None]), str(out))
    assert not (out / "b.npy").exists()
    with pytest.raises(NotImplementedError):
        D.get_input_file_list("gs://bucket/dir")
