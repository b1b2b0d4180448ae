#!/usr/bin/env python
"""Generates tests/golden/jukebox_full36.npz: the CPU oracle run ONCE, in the build container, on the exact
configuration bench.py times (BASELINE configs[1]): synthetic clip 0 (25 s @ 44.1 kHz) -> VQ-VAE level-2 codes ->
ALL 36 prior layers at 5b widths -> (240, 4800) pooled embedding (jukebox/main.py:54-68,101-110,157-167).

    python tests/golden/make_jukebox_full_golden.py          # ~10 min on 8 vCPUs, ~12 GB RAM

Stored: codes (C oracle = the defined-order bit-exact restatement), the codes of the order-free torch F.conv1d
restatement and their agreement, the best / second-best codebook distance gap per token (float64; the near-tie
audit of SURVEY section 7 "hard parts"), probe rows of the un-pooled residual stream at several depths, the pooled
embeddings (f = 10 -> (240,4800); f = 0 -> (4800,)), a checksum of the data-dependent codebook, and the oracle's
own max|acts| per probed depth.  Weights are NOT stored: tests/fulldepth.py regenerates them from the CPU seed.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import fulldepth as FD  # noqa: E402
from oracle import jukebox_c as C  # noqa: E402
from oracle import jukebox_ref as R  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    hps = FD.jukebox_hps()
    t0 = time.time()
    w = FD.jukebox_weights_cpu(hps)
    print(f"weights: {time.time() - t0:.1f}s", flush=True)
    # data-dependent codebook from the calibration clip (bench.py does this through the HIP encoder, which is
    # bit-exact with the C oracle: tests/test_vqvae_gpu.py)
    xe_cal = C.encoder_forward(w, FD.jukebox_clip(FD.CAL_CLIP, hps)[None], hps)
    k = FD.codebook_from_encoding(xe_cal, hps)
    w["bottleneck.level_blocks.2.k"] = k
    a = FD.jukebox_clip(FD.GOLD_CLIP, hps)
    t0 = time.time()
    codes, enc, mind = C.encode_codes(w, a[None], hps, return_all=True)
    print(f"C-oracle encode: {time.time() - t0:.1f}s", flush=True)
    # order-free restatement (torch F.conv1d + the upstream distance formula) and the near-tie audit
    with torch.no_grad():
        xe_t = R.vqvae_encoder_forward(w, torch.from_numpy(a)[None, None], hps)
        codes_t = R.bottleneck_encode(k, xe_t)[0].numpy()
    x64 = enc[0].T.astype(np.float64)                                   # (8192, 64)
    k64 = k.numpy().astype(np.float64)
    d = (x64 ** 2).sum(1, keepdims=True) - 2 * x64 @ k64.T + (k64 ** 2).sum(1)[None]
    part = np.partition(d, 1, axis=1)
    gap = (part[:, 1] - part[:, 0]).astype(np.float32)
    best64 = d.argmin(1)
    agree_torch = float((codes_t == codes[0]).mean())
    agree_f64 = float((best64 == codes[0]).mean())
    enc_diff = float(np.abs(xe_t[0].numpy() - enc[0]).max())
    print(f"codes: torch-vs-C agreement {agree_torch:.5f}, float64-argmin-vs-C {agree_f64:.5f}; "
          f"encoder max|torch - C| = {enc_diff:.3e}; min gap {gap.min():.3e}, median {np.median(gap):.3e}", flush=True)

    z = torch.from_numpy(codes)
    x_cond, y_cond = R.get_cond(w, hps)
    h = R.prior_embed(w, z, x_cond, y_cond, hps)
    probes, maxabs = {}, {}
    rows = list(FD.PROBE_ROWS)
    t0 = time.time()
    with torch.no_grad():
        for dl in range(hps.prior_depth):
            h = R.prior_layer(w, h, dl, hps)
            if dl + 1 in FD.PROBE_LAYERS:
                probes[dl + 1] = h[0, rows].numpy().copy()
                maxabs[dl + 1] = float(h.abs().max())
            print(f"layer {dl + 1}/{hps.prior_depth}  {time.time() - t0:.0f}s  max|h| {float(h.abs().max()):.3f}", flush=True)
    acts = h[0].float()
    frame_len = int(np.floor((hps.n_ctx / (hps.sample_length / hps.sr)) / 10))
    pooled = R.windowed_average(acts, frame_len)[0].numpy()
    assert pooled.shape == (240, hps.prior_width)
    out = dict(
        codes=codes[0].astype(np.int16), codes_torch=codes_t.astype(np.int16), gap=gap,
        agree_torch=np.float64(agree_torch), agree_f64=np.float64(agree_f64), enc_maxdiff_torch_vs_c=np.float64(enc_diff),
        codebook_sha=np.array(FD.sha(k.numpy())), audio_sha=np.array(FD.sha(a)),
        probe_rows=np.array(rows), probe_layers=np.array(sorted(probes)),
        probes=np.stack([probes[l] for l in sorted(probes)]), maxabs=np.array([maxabs[l] for l in sorted(probes)]),
        emb_f10=pooled.astype(np.float32), emb_f0=acts.mean(0).numpy().astype(np.float32),
        acts_maxabs=np.float64(acts.abs().max()),
    )
    np.savez_compressed(FD.JUKEBOX_NPZ, **out)
    print("wrote", FD.JUKEBOX_NPZ, os.path.getsize(FD.JUKEBOX_NPZ) / 1e6, "MB")


if __name__ == "__main__":
    main()
