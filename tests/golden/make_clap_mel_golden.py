#!/usr/bin/env python
"""Generates tests/golden/clap_mel.npz with transformers.ClapFeatureExtractor (non-fusion path: rand_trunc / repeatpad, slaney
filter bank -- the published algorithm of laion_clap's front end): a 2.3 s synthetic music-like clip (decaying harmonics + noise,
~70 dB of spectral dynamic range) repeat-padded to 10 s.  Usage: python tests/golden/make_clap_mel_golden.py"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from transformers import ClapFeatureExtractor  # noqa: E402


def synth(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 48000.0
    x = sum(0.3 / (h ** 1.5) * np.sin(2 * np.pi * 196.0 * h * t + rng.uniform(0, 6.28)) for h in range(1, 30))
    x = x * np.exp(-1.5 * (t % 0.5)) + 1e-3 * rng.standard_normal(n)
    return (0.5 * x / np.abs(x).max()).astype(np.float32)


def main():
    fe = ClapFeatureExtractor(feature_size=64, sampling_rate=48000, hop_length=480, max_length_s=10, fft_window_size=1024,
                              frequency_min=50, frequency_max=14000, truncation="rand_trunc", padding="repeatpad")
    wave = synth(110400, 3)
    feats = np.asarray(fe(raw_speech=wave, sampling_rate=48000, return_tensors="np")["input_features"])[0, 0]
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clap_mel.npz")
    np.savez_compressed(dst, wave=wave, logmel=feats.astype(np.float32), filters=np.asarray(fe.mel_filters_slaney, np.float64))
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB", feats.shape, feats.min(), feats.max())


if __name__ == "__main__":
    main()
