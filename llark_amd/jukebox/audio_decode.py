"""File decoding of ``load_audio_from_file`` (jukebox/main.py:31): what ``librosa.load`` gets from libsndfile for the containers it
reads without further codecs -- RIFF wav (``scipy.io.wavfile`` + soundfile's integer scaling), FLAC (``llark_flac_decode_host``: a
native decoder whose output is checked against the stream's own MD5 signature), and the plain-PCM containers AIFF / AIFF-C
(``NONE`` / ``sowt`` / ``fl32`` / ``fl64``) and Sun / NeXT ``.au`` -- as ``(sample_rate, float32 [frames] or [frames][channels])``.
Anything else (mp3 / ogg: librosa's audioread fallback, which needs ffmpeg or gstreamer codecs) is not decoded."""
from __future__ import annotations

import ctypes
import io
from typing import Tuple

import numpy as np



class UnsupportedContainerError(RuntimeError):
    """The bytes are not one of the containers decoded here.  NOT a ValueError on purpose: load_audio_from_file turns ValueError into
    EmptyFileError ("probably empty", jukebox/main.py:29-34) and its callers skip such files silently -- an mp3 must not be reported
    as an empty file."""

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


def _pcm_to_float(raw: bytes, bits: int, channels: int, big_endian: bool, is_float: bool = False, signed8: bool = True) -> np.ndarray:
    """Interleaved PCM bytes -> float32 [frames][channels] with libsndfile's scaling (integers / 2^(bits - 1))."""
    e = ">" if big_endian else "<"
    if is_float:
        x = np.frombuffer(raw, dtype=e + ("f4" if bits == 32 else "f8")).astype(np.float32)
    elif bits == 8:
        x = np.frombuffer(raw, dtype=np.int8 if signed8 else np.uint8).astype(np.float32)
        x = x / 128.0 if signed8 else (x - 128.0) / 128.0
    elif bits == 16:
        x = np.frombuffer(raw, dtype=e + "i2").astype(np.float32) / 32768.0
    elif bits == 24:
        b = np.frombuffer(raw[: len(raw) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        hi, mid, lo = (b[:, 0], b[:, 1], b[:, 2]) if big_endian else (b[:, 2], b[:, 1], b[:, 0])
        v = (hi << 16) | (mid << 8) | lo
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif bits == 32:
        x = np.frombuffer(raw, dtype=e + "i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported PCM sample size {bits}")
    n = x.shape[0] // channels
    return x[: n * channels].reshape(n, channels)


def _ieee_extended(b: bytes) -> float:
    """The 80-bit extended float AIFF stores its sample rate in."""
    exp = ((b[0] & 0x7F) << 8) | b[1]
    mant = int.from_bytes(b[2:10], "big")
    if exp == 0 and mant == 0:
        return 0.0
    return (-1.0 if b[0] & 0x80 else 1.0) * mant * 2.0 ** (exp - 16383 - 63)


def decode_aiff(data: bytes) -> Tuple[int, np.ndarray]:
    """AIFF / AIFF-C: big-endian chunks; ``COMM`` (channels, frames, sample size, 80-bit rate[, compression]) + ``SSND`` (offset,
    block size, samples).  Compression ``NONE`` / ``twos`` (big-endian PCM), ``sowt`` (little-endian), ``fl32`` / ``fl64``."""
    if len(data) < 12 or data[:4] != b"FORM" or data[8:12] not in (b"AIFF", b"AIFC"):
        raise ValueError("aiff: no FORM/AIFF header")
    aifc = data[8:12] == b"AIFC"
    p, comm, ssnd = 12, None, None
    while p + 8 <= len(data):
        cid, size = data[p:p + 4], int.from_bytes(data[p + 4:p + 8], "big")
        body = data[p + 8:p + 8 + size]
        if cid == b"COMM":
            comm = body
        elif cid == b"SSND":
            ssnd = body
        p += 8 + size + (size & 1)
    if comm is None or len(comm) < 18:
        raise ValueError("aiff: no COMM chunk")
    channels, frames, bits = int.from_bytes(comm[0:2], "big"), int.from_bytes(comm[2:6], "big"), int.from_bytes(comm[6:8], "big")
    sr = int(round(_ieee_extended(comm[8:18])))
    comp = comm[18:22] if aifc and len(comm) >= 22 else b"NONE"
    if comp not in (b"NONE", b"twos", b"sowt", b"fl32", b"FL32", b"fl64", b"FL64"):
        raise ValueError(f"aiff: compression {comp!r} is not decoded")
    if channels < 1 or sr <= 0:
        raise ValueError("aiff: bad COMM chunk")
    if ssnd is None or frames == 0:
        return sr, np.zeros((0, channels), np.float32)
    off = int.from_bytes(ssnd[0:4], "big")
    raw = ssnd[8 + off:]
    is_float = comp.lower() in (b"fl32", b"fl64")
    width = 64 if comp.lower() == b"fl64" else 32 if is_float else (bits + 7) // 8 * 8
    x = _pcm_to_float(raw[: frames * channels * width // 8], width, channels, big_endian=comp != b"sowt", is_float=is_float)
    return sr, x[:frames]


def decode_au(data: bytes) -> Tuple[int, np.ndarray]:
    """Sun / NeXT .au: ``.snd``, header size, data size, encoding (2..7 = 8 / 16 / 24 / 32-bit PCM, float32, float64), rate, channels;
    big-endian."""
    if len(data) < 24 or data[:4] != b".snd":
        raise ValueError("au: no .snd header")
    hdr, size, enc, sr, channels = (int.from_bytes(data[i:i + 4], "big") for i in (4, 8, 12, 16, 20))
    fmt = {2: (8, False), 3: (16, False), 4: (24, False), 5: (32, False), 6: (32, True), 7: (64, True)}.get(enc)
    if fmt is None:
        raise ValueError(f"au: encoding {enc} is not decoded")
    if channels < 1 or sr <= 0 or hdr < 24:
        raise ValueError("au: bad header")
    raw = data[hdr:] if size in (0xFFFFFFFF, 0) else data[hdr:hdr + size]
    return sr, _pcm_to_float(raw, fmt[0], channels, big_endian=True, is_float=fmt[1])


def decode_audio(f) -> Tuple[int, np.ndarray]:
    """Path or binary file object -> (sample_rate, float32 [frames] or [frames][channels]); the container is told from its first bytes."""
    data = _read_all(f)
    if data[:3] == b"ID3" and len(data) >= 10:          # ID3v2 tag in front of the stream (almost always mp3, sometimes FLAC): skip it, sniff what follows
        size = ((data[6] & 0x7F) << 21) | ((data[7] & 0x7F) << 14) | ((data[8] & 0x7F) << 7) | (data[9] & 0x7F)
        body = data[10 + size + (10 if data[5] & 0x10 else 0):]
        if body[:4] != b"fLaC":
            raise UnsupportedContainerError(f"unsupported audio container: ID3v2 tag followed by {body[:4]!r} (mp3 is not decoded): "
                                            "wav, FLAC, AIFF and .au are")
        data = body
    head = data[:4]
    if head == b"fLaC":
        return decode_flac(data)
    if head in (b"RIFF", b"RIFX", b"RF64") or len(data) == 0:
        return decode_wav(io.BytesIO(data))
    if head == b"FORM":
        return decode_aiff(data)
    if head == b".snd":
        return decode_au(data)
    raise UnsupportedContainerError(f"unsupported audio container (first bytes {head!r}): wav, FLAC, AIFF and .au are decoded")
