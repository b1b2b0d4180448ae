"""File decoding of ``load_audio_from_file`` (jukebox/main.py:31): what ``librosa.load`` gets from libsndfile for the two containers
handled here -- RIFF wav (``scipy.io.wavfile`` + soundfile's integer scaling) and FLAC (``llark_flac_decode_host``: a native decoder
whose output is checked against the stream's own MD5 signature) -- as ``(sample_rate, float32 [frames] or [frames][channels])``.
Anything else (mp3 / ogg: librosa's audioread fallback, which needs ffmpeg or gstreamer codecs) is not decoded."""
from __future__ import annotations

import ctypes
import io
from typing import Tuple

import numpy as np


def _read_all(f) -> bytes:
    if hasattr(f, "read"):
        return f.read()
    with open(f, "rb") as fh:
        return fh.read()


def decode_flac(data: bytes, verify_md5: bool = True) -> Tuple[int, np.ndarray]:
    """FLAC bytes -> (sample_rate, float32 [frames][channels]) scaled by 2^-(bits - 1) like ``soundfile.read(dtype="float32")``."""
    from .. import _lib

    L = _lib.lib()
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    sr, ch, bps, total = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()

    def chk(rc, what):
        if rc != 0:
            raise ValueError(f"{what}: {L.llark_last_error().decode()}")

    chk(L.llark_flac_info_host(buf, len(data), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(bps), ctypes.byref(total)), "flac")
    frames = ctypes.c_int64(total.value)
    if total.value == 0:                                  # STREAMINFO does not record the length: count first
        chk(L.llark_flac_decode_host(buf, len(data), None, 0, ctypes.byref(frames), 0), "flac")
    out = np.empty((max(frames.value, 0), ch.value), dtype=np.int32)
    got = ctypes.c_int64()
    chk(L.llark_flac_decode_host(buf, len(data), out.ctypes.data_as(ctypes.c_void_p), out.shape[0], ctypes.byref(got), int(verify_md5)), "flac")
    return sr.value, (out.astype(np.float32) / np.float32(2.0 ** (bps.value - 1)))


def decode_wav(f) -> Tuple[int, np.ndarray]:
    """RIFF wav -> (sample_rate, float32) with soundfile's scaling: int16 / 2^15, int32 (and 24-bit in int32) / 2^31, uint8 -> (x - 128) / 2^7."""
    from scipy.io import wavfile

    sr, data = wavfile.read(f)
    if data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    return int(sr), data.astype(np.float32)


def decode_audio(f) -> Tuple[int, np.ndarray]:
    """Path or binary file object -> (sample_rate, float32 [frames] or [frames][channels]); the container is told from its first bytes."""
    data = _read_all(f)
    head = data[:4]
    if head == b"fLaC" or head[:3] == b"ID3":
        return decode_flac(data)
    if head in (b"RIFF", b"RIFX", b"RF64") or len(data) == 0:
        return decode_wav(io.BytesIO(data))
    raise ValueError(f"unsupported audio container (first bytes {head!r}): wav and FLAC are decoded")
