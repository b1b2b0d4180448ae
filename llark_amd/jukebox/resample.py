"""Sample-rate conversion of ``load_audio_from_file`` (jukebox/main.py:31, ``lr.load(fpath, sr=44100)``).

Which resampler "the reference" is depends on the librosa it runs under, and the repository pins none
(requirements.txt:2).  Two are restated here, neither can be pinned offline (no librosa / resampy / soxr in this image):

* ``"kaiser_best"`` (default) -- what the image of jukebox/main.py actually runs: docker/jukebox-embed.dockerfile builds from
  python 3.7 and ``pip install -e jukebox`` (openai/jukebox @ 08efbbc pins ``librosa==0.7.2``), whose ``load`` resamples with
  resampy's ``kaiser_best`` filter: band-limited sinc interpolation over a table of 64 zero crossings x 512 samples,
  Kaiser window beta = 14.769656459379492, roll-off 0.9475937167399596 (resampy/filters.py documents the parameters; the
  table is regenerated from them here, the interpolation loop is ``llark_resample_sinc_host``); output length
  ``int(n * ratio)`` fixed up to ``ceil(n * ratio)`` as ``librosa.core.resample`` does.
* ``"soxr_hq"`` -- librosa >= 0.10's default.  libsoxr's algorithm (multi-stage, FFT-based) is not restated; what is kept is
  its published HQ specification -- linear phase, pass band to 0.913 of the lower Nyquist rate, stop band from 1.0, 20-bit
  (120 dB) rejection -- realised as ONE Kaiser-windowed polyphase FIR.  The two agree wherever both are transparent.

Both are within 1e-5 of the ideal band-limited interpolation of an in-band signal (tests/test_audio_io_cpu.py); they differ from
each other, and from the packages they restate, in the transition band and at the -100 dB level.  44.1 kHz input is not touched.
"""
from __future__ import annotations

import ctypes
from fractions import Fraction
from functools import lru_cache
from math import ceil

import numpy as np

KAISER_BEST = dict(num_zeros=64, precision=9, beta=14.769656459379492, rolloff=0.9475937167399596)
SOXR_HQ = dict(passband_end=0.913, stopband_begin=1.0, rejection_db=125.0)     # 20-bit precision + margin for the window design
RES_TYPES = ("kaiser_best", "soxr_hq")


@lru_cache(maxsize=2)
def sinc_window(num_zeros: int, precision: int, beta: float, rolloff: float):
    """Right half of resampy's interpolation window (resampy/filters.py ``sinc_window``): ``rolloff * sinc(rolloff * t)`` for
    t in [0, num_zeros] sampled ``2**precision`` times per zero crossing, tapered by the right half of a symmetric Kaiser
    window.  Returns (half_window float64 [num_zeros * 2**precision + 1], samples per zero crossing)."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return (taper * sinc_win).astype(np.float64), num_bits


def resample_kaiser_best(x: np.ndarray, sr_orig: int, sr_new: int) -> np.ndarray:
    """``librosa.core.resample(y, orig_sr, target_sr, res_type="kaiser_best")`` of librosa 0.7.2 for mono float32 input."""
    from .. import _lib

    x = np.ascontiguousarray(x, dtype=np.float32)
    ratio = float(sr_new) / float(sr_orig)
    n_out = int(x.shape[0] * ratio)                       # resampy.resample: shape[axis] = int(shape[axis] * sample_ratio)
    if n_out < 1:
        raise ValueError(f"Input signal length={x.shape[0]} is too small to resample from {sr_orig}->{sr_new}")
    half, num_table = sinc_window(**KAISER_BEST)
    win = half * ratio if ratio < 1 else half.copy()
    dwin = np.zeros_like(win)
    dwin[:-1] = np.diff(win)
    y = np.empty((n_out,), dtype=np.float32)
    L = _lib.lib()
    rc = L.llark_resample_sinc_host(x.ctypes.data_as(ctypes.c_void_p), x.shape[0], ratio, win.ctypes.data_as(ctypes.c_void_p),
                                    dwin.ctypes.data_as(ctypes.c_void_p), win.shape[0], num_table, y.ctypes.data_as(ctypes.c_void_p), n_out)
    if rc != 0:
        raise _lib.LlarkHipError(f"llark_resample_sinc_host failed ({rc}): {L.llark_last_error().decode()}")
    n_fix = int(ceil(x.shape[0] * ratio))                 # librosa.core.resample: util.fix_length(y_hat, n_samples)
    if n_fix > n_out:
        y = np.pad(y, (0, n_fix - n_out))
    return y


@lru_cache(maxsize=8)
def _soxr_hq_taps(up: int, down: int):
    from scipy.signal import firwin, kaiser_beta

    fn = 1.0 / max(up, down)                              # the lower Nyquist rate, in units of the Nyquist rate of the up-sampled grid
    width = (SOXR_HQ["stopband_begin"] - SOXR_HQ["passband_end"]) * fn
    a = SOXR_HQ["rejection_db"]
    numtaps = int(ceil((a - 7.95) / (2.285 * np.pi * width))) + 1
    numtaps += 1 - numtaps % 2                            # odd length: linear phase with an integer delay
    cutoff = 0.5 * (SOXR_HQ["stopband_begin"] + SOXR_HQ["passband_end"]) * fn
    return firwin(numtaps, cutoff, window=("kaiser", kaiser_beta(a)))          # unit DC gain; resample_poly multiplies by `up`


def resample_soxr_hq_spec(x: np.ndarray, sr_orig: int, sr_new: int) -> np.ndarray:
    """A polyphase FIR to libsoxr's HQ specification (see the module docstring); length ``ceil(n * ratio)`` like librosa."""
    from scipy.signal import resample_poly

    fr = Fraction(int(sr_new), int(sr_orig))
    y = resample_poly(np.asarray(x, dtype=np.float64), fr.numerator, fr.denominator, window=_soxr_hq_taps(fr.numerator, fr.denominator))
    return y.astype(np.float32)


def resample(x: np.ndarray, sr_orig: int, sr_new: int, res_type: str = "kaiser_best") -> np.ndarray:
    if sr_orig <= 0 or sr_new <= 0:
        raise ValueError(f"Invalid sample rate: {sr_orig} -> {sr_new}")
    if int(sr_orig) == int(sr_new):
        return np.asarray(x, dtype=np.float32)
    if res_type == "kaiser_best":
        return resample_kaiser_best(x, sr_orig, sr_new)
    if res_type == "soxr_hq":
        return resample_soxr_hq_spec(x, sr_orig, sr_new)
    raise ValueError(f"res_type must be one of {RES_TYPES}, got {res_type!r}")
