"""``llark_flac_decode_host`` (csrc/audio_host.hip) against streams written by tests/flac_writer.py: every subframe type, residual
form, stereo decorrelation and header variant decodes to the integers that were encoded; the stream's MD5 signature, CRC-8 and
CRC-16 are enforced; ``load_audio_from_file`` reads FLAC like wav (jukebox/main.py:29-45 through libsndfile).  No FLAC file or
encoder exists in this image: the writer follows the published format, the decoder is checked against it and against the
format's own self-certification (MD5 of the decoded samples)."""
import io

import numpy as np
import pytest

import flac_writer as FW
from llark_amd.jukebox import audio_decode as AD
from llark_amd.jukebox import extract as E


def _music(n, ch, bps, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    amp = 2 ** (bps - 1) * 0.4
    x = np.stack([amp * np.sin(2 * np.pi * (220.0 * (c + 1)) * t / 44100 + c) + amp * 0.02 * rng.standard_normal(n) for c in range(ch)], axis=1)
    return np.round(x).astype(np.int64)


def _decode(data, bps):
    sr, x = AD.decode_flac(data)
    return sr, np.round(x.astype(np.float64) * 2.0 ** (bps - 1)).astype(np.int64)


def _decode_int32(data):
    """llark_flac_decode_host's raw output: interleaved int32 [frames][channels]."""
    import ctypes

    from llark_amd import _lib
    L = _lib.lib()
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    sr, ch, bps, total, got = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64()
    assert L.llark_flac_info_host(buf, len(data), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(bps), ctypes.byref(total)) == 0
    out = np.empty((total.value, ch.value), dtype=np.int32)
    assert L.llark_flac_decode_host(buf, len(data), out.ctypes.data_as(ctypes.c_void_p), total.value, ctypes.byref(got), 1) == 0, L.llark_last_error()
    assert got.value == total.value
    return out.astype(np.int64)


@pytest.mark.parametrize("bps", [8, 16, 24])
@pytest.mark.parametrize("stereo", ["independent", "left_side", "right_side", "mid_side"])
def test_stereo_modes_and_sample_sizes(bps, stereo):
    x = _music(5000, 2, bps, seed=bps)
    data = FW.write_flac(x, 44100, bps, block=1152, stereo=stereo, sub=lambda c: dict(kind="fixed", order=3, porder=3))
    sr, got = _decode(data, bps)
    assert sr == 44100 and got.shape == x.shape
    np.testing.assert_array_equal(got, x)


@pytest.mark.parametrize("sub", [
    dict(kind="verbatim"),
    dict(kind="fixed", order=0, porder=0),
    dict(kind="fixed", order=1, porder=1, method=1),
    dict(kind="fixed", order=4, porder=4),
    dict(kind="fixed", order=2, porder=2, escape_partition=1),
    dict(kind="lpc", lpc=[1946, -973], lpc_prec=12, lpc_shift=10, porder=2),
    dict(kind="lpc", lpc=[3, -3, 1, 0, 0, 0, 0, 1, -1], lpc_prec=5, lpc_shift=0, porder=0, method=1),
])
def test_subframe_types_and_residual_forms(sub):
    x = _music(4096 + 100, 1, 16, seed=3)                               # a shorter last block with an explicit 16-bit block size
    data = FW.write_flac(x, 48000, 16, block=4096, sub=lambda c: dict(sub))
    sr, got = _decode(data, 16)
    assert sr == 48000
    np.testing.assert_array_equal(got, x)


def test_constant_wasted_bits_odd_rates_small_blocks_and_metadata():
    n = 16 * 300                                                         # 300 frames: the coded frame number needs two bytes
    x = _music(n, 1, 16, seed=5)
    x = (x >> 3) << 3                                                    # three wasted bits in every subframe
    x[160:176] = 1234 * 8                                                # one block of a constant
    subs = lambda c: dict(kind="fixed", order=1, porder=0)
    data = FW.write_flac(x, 37800, 16, block=16, sub=subs, extra_blocks=[b"vendor=test" * 3], id3=True)
    sr, got = _decode(data, 16)
    assert sr == 37800                                                   # not in the rate table: 16-bit Hz field in every frame header
    np.testing.assert_array_equal(got, x)
    # constant subframes, header fields taken from STREAMINFO, total length not recorded (decoder counts first)
    z = np.full((1000, 2), -77, dtype=np.int64)
    data = FW.write_flac(z, 44100, 16, block=500, sub=lambda c: dict(kind="constant"), total_known=False, header_from_streaminfo=True)
    sr, got = _decode(data, 16)
    np.testing.assert_array_equal(got, z)
    # 32-bit samples with a 33-bit side channel
    big = _music(600, 2, 32, seed=9)
    data = FW.write_flac(big, 96000, 32, block=192, stereo="mid_side", sub=lambda c: dict(kind="verbatim"))
    sr, got = _decode(data, 32)
    assert sr == 96000
    assert int(np.abs(got - big).max()) <= 64                            # float32 carries 24 of the 32 bits (ulp 64 at 0.4 x 2^31) ...
    np.testing.assert_array_equal(_decode_int32(data), big)              # ... the decoder's own int32 output is exact


def test_corruption_is_detected():
    x = _music(3000, 2, 16, seed=11)
    good = FW.write_flac(x, 44100, 16, block=1024)
    AD.decode_flac(good)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x10                                           # inside a frame: CRC-16 (or a structural check) must trip
    with pytest.raises(ValueError, match="flac"):
        AD.decode_flac(bytes(bad))
    wrong_md5 = bytearray(good)
    wrong_md5[4 + 4 + 18 + 3] ^= 0xFF                                    # a byte of the MD5 signature in STREAMINFO
    with pytest.raises(ValueError, match="MD5"):
        AD.decode_flac(bytes(wrong_md5))
    assert AD.decode_flac(bytes(wrong_md5), verify_md5=False)[1].shape == (3000, 2)
    with pytest.raises(ValueError, match="flac"):
        AD.decode_flac(good[: len(good) - 40])                           # truncated: fewer samples than STREAMINFO records
    with pytest.raises(ValueError, match="fLaC"):
        AD.decode_flac(b"OggS" + bytes(100))
    no_md5 = FW.write_flac(x, 44100, 16, block=1024, with_md5=False)     # signature all zero = not computed: accepted
    np.testing.assert_array_equal(_decode(no_md5, 16)[1], x)


def test_load_audio_from_file_reads_flac_like_wav(tmp_path):
    from scipy.io import wavfile

    x = _music(30000, 2, 16, seed=2)
    pw, pf = tmp_path / "a.wav", tmp_path / "a.flac"
    wavfile.write(pw, 44100, x.astype(np.int16))
    pf.write_bytes(FW.write_flac(x, 44100, 16, block=4096, stereo="mid_side"))
    a, b = E.load_audio_from_file(str(pw)), E.load_audio_from_file(str(pf))
    np.testing.assert_array_equal(a, b)                                  # same samples, same scaling, same mono mean and peak normalisation
    np.testing.assert_array_equal(E.load_audio_from_file(io.BytesIO(pf.read_bytes())), a)
    x22 = _music(22050, 1, 16, seed=4)
    (tmp_path / "lo.flac").write_bytes(FW.write_flac(x22, 22050, 16))
    wavfile.write(tmp_path / "lo.wav", 22050, x22[:, 0].astype(np.int16))
    np.testing.assert_array_equal(E.load_audio_from_file(tmp_path / "lo.flac"), E.load_audio_from_file(tmp_path / "lo.wav"))
    (tmp_path / "x.ogg").write_bytes(b"OggS" + bytes(64))
    with pytest.raises(AD.UnsupportedContainerError):                    # a container that is not decoded here is NOT reported as an empty file
        E.load_audio_from_file(tmp_path / "x.ogg")
