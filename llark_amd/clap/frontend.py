"""Waveform side of the CLAP plugin: the host half of ``scripts/clap/clap_embeddings.py:127-153`` (``load_audio_input``:
int16 round trip, ``get_audio_features`` with ``data_truncating="rand_trunc"`` / ``data_filling="repeatpad"`` to 480000
samples) and the spectrogram extractor that laion_clap runs inside the model, here one fused HIP kernel
(``llark_clap_logmel``).  The tables the kernel needs (periodic hann window, FFT twiddles, slaney mel filters) are
load-time constants built in float64 numpy."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from .. import ops as O

SAMPLE_RATE, N_FFT, HOP, CLIP_SAMPLES, N_MELS, F_MIN, F_MAX = 48000, 1024, 480, 480000, 64, 50.0, 14000.0


def slaney_mel_filters(sr: int = SAMPLE_RATE, n_fft: int = N_FFT, n_mels: int = N_MELS, fmin: float = F_MIN, fmax: float = F_MAX) -> np.ndarray:
    """The filter bank torchlibrosa's LogmelFilterBank takes from librosa.filters.mel (slaney scale: linear below 1 kHz,
    log above; area-normalised triangles): fp32 (n_mels, n_fft // 2 + 1)."""
    lin_step, knee_hz = 200.0 / 3.0, 1000.0
    knee_mel, log_step = knee_hz / lin_step, np.log(6.4) / 27.0
    to_mel = lambda hz: knee_mel + np.log(hz / knee_hz) / log_step if hz >= knee_hz else hz / lin_step
    mels = np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2)
    edges = np.where(mels >= knee_mel, knee_hz * np.exp(log_step * (mels - knee_mel)), lin_step * mels)     # band edges in Hz
    bins = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    bank = np.zeros((n_mels, bins.size))
    for m in range(n_mels):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        rise, fall = (bins - lo) / (mid - lo), (hi - bins) / (hi - mid)
        bank[m] = np.maximum(0.0, np.minimum(rise, fall)) * (2.0 / (hi - lo))
    return bank.astype(np.float32)


def fit_clip(wave: np.ndarray, rng: Optional[np.random.Generator] = None, max_len: int = CLIP_SAMPLES) -> np.ndarray:
    """rand_trunc: a random max_len crop of a longer clip; repeatpad: whole repeats of a shorter one, then zeros."""
    wave = np.asarray(wave, np.float32).reshape(-1)
    if wave.size == 0:
        raise ValueError("empty waveform")
    if wave.size > max_len:
        start = int((rng or np.random.default_rng()).integers(0, wave.size - max_len + 1))
        return wave[start:start + max_len]
    if wave.size < max_len:
        wave = np.tile(wave, max_len // wave.size)
        wave = np.concatenate([wave, np.zeros(max_len - wave.size, np.float32)])
    return wave


class ClapFrontend:
    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        k = np.arange(N_FFT)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * k / N_FFT)
        ang = -2.0 * np.pi * np.arange(N_FFT // 2) / N_FFT
        bank = slaney_mel_filters()
        nz = bank > 0
        lo = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
        hi = np.where(nz.any(1), bank.shape[1] - nz[:, ::-1].argmax(1), 0).astype(np.int32)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.window = dev(window.astype(np.float32))
        self.twiddle = dev(np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32))
        self.melw, self.mel_lo, self.mel_hi = dev(bank), dev(lo), dev(hi)

    def logmel(self, wav: torch.Tensor, quantize_int16: bool = True) -> torch.Tensor:
        """(B, n) fp32 waveform on the GPU -> (B, 1, n // 480 + 1, 64) log-mel dB."""
        if wav.dim() != 2:
            raise ValueError(f"expected (B, samples) waveforms, got {tuple(wav.shape)}")
        return O.clap_logmel(wav.to(torch.float32).contiguous(), self.window, self.twiddle, self.melw, self.mel_lo, self.mel_hi,
                             quantize_int16).unsqueeze(1)

    def batch(self, waves: Sequence[np.ndarray], rng: Optional[np.random.Generator] = None) -> torch.Tensor:
        """Host clips of any length -> (B, 480000) on the GPU (rand_trunc / repeatpad)."""
        return torch.from_numpy(np.stack([fit_clip(w, rng) for w in waves])).to(self.device)
