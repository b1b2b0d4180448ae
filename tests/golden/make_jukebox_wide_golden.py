#!/usr/bin/env python
"""Generates tests/golden/jukebox_full36_wide.npz (round 4; VERDICT r03 items 1 + 2): the CPU oracle run in the build
container on MORE of the configuration bench.py times than tests/golden/jukebox_full36.npz (one clip) holds.

    python tests/golden/make_jukebox_wide_golden.py [stage ...]      # stages: codes rich short12 outlier head64; default all
                                                                      # ~50 min on 8 vCPUs, ~14 GB RAM; saves after every stage

Contents (weights are NOT stored: tests/fulldepth.py regenerates them from the CPU seed on both sides):

* ``codes10`` -- the C oracle's VQ codes (oracle/jukebox_ref.c: the defined-order, bit-exact restatement) of clips 0..9 =
  the 8 clips bench.py encodes on rank 0 + 2 more, with the best / second-best codebook distance gap of every token
  (float64) summarised per clip (the 64 smallest gaps and their token indices): the table the default (fused) HIP
  encoder has to reproduce with 0 mismatches.
* three more full-depth cases (36 layers at 5b widths, fp32 torch oracle = oracle/jukebox_ref.py), each with its codes,
  probe rows of the residual stream at depth 1/3/6/12/24/36, the pooled embedding (f = 10) and the global mean (f = 0):
    ``rich``     a clip with a different spectrum (llark_amd.jukebox.synthetic.synthetic_clip_rich)
    ``short12``  a 12 s clip: latent_audio_len = 4134 rows survive the slice of jukebox/main.py:154 -> 121 frames, not 240
    ``outlier``  clip 0 through weights whose prior layers carry x30 outlier channels (tests/fulldepth.add_outlier_channels)
* ``head64_*`` -- the SAME graph evaluated in float64 over the first 1024 tokens (every attention pattern of the prior is
  causal, so a prefix is self-contained) for the base clip and the three cases: a noise-floor reference that tells how much of
  an fp32-vs-HIP difference is the fp32 oracle's own rounding.
"""
from __future__ import annotations

import math
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

OUT = {}


def save():
    np.savez_compressed(FD.WIDE_NPZ, **OUT)
    print("saved", FD.WIDE_NPZ, f"{os.path.getsize(FD.WIDE_NPZ) / 1e6:.2f} MB, keys: {len(OUT)}", flush=True)


def pad(a, hps):
    return np.pad(a, (0, max(0, hps.sample_length - len(a))))[: hps.sample_length].astype(np.float32)


def gaps64(enc, k):
    x64 = enc.T.astype(np.float64)
    k64 = k.numpy().astype(np.float64)
    d = (x64 ** 2).sum(1, keepdims=True) - 2 * x64 @ k64.T + (k64 ** 2).sum(1)[None]
    part = np.partition(d, 1, axis=1)
    return (part[:, 1] - part[:, 0]), part[:, 0], d.argmin(1)


def frame_len(hps):
    return int(math.floor((hps.n_ctx / (hps.sample_length / hps.sr)) / 10))


def stage_codes(w, hps):
    codes, small_gap, small_tok, agree = [], [], [], []
    for i in FD.CODE_CLIPS:
        t0 = time.time()
        a = FD.jukebox_clip(i, hps)
        c, enc, _ = C.encode_codes(w, a[None], hps, return_all=True)
        gap, _best, arg64 = gaps64(enc[0], w["bottleneck.level_blocks.2.k"])
        order = np.argsort(gap)[:64]
        codes.append(c[0].astype(np.int16))
        small_gap.append(gap[order].astype(np.float32))
        small_tok.append(order.astype(np.int16))
        agree.append(float((arg64 == c[0]).mean()))
        print(f"codes clip {i}: {time.time() - t0:.1f}s  min gap {gap.min():.3e}  float64-argmin agreement {agree[-1]:.5f}", flush=True)
    OUT.update(codes10=np.stack(codes), codes10_clips=np.array(FD.CODE_CLIPS), codes10_small_gap=np.stack(small_gap),
               codes10_small_tok=np.stack(small_tok), codes10_agree_f64=np.array(agree))


def run_layers(w, hps, codes, dtype, tokens=None, tag=""):
    """codes (n_ctx,) -> (acts (tokens, width) in dtype, probes dict) through all prior layers."""
    z = torch.from_numpy(codes.astype(np.int64))[None]
    x_cond, y_cond = R.get_cond(w, hps)
    h = R.prior_embed(w, z, x_cond, y_cond, hps)
    if tokens is not None:
        h = h[:, :tokens].contiguous()
    h = h.to(dtype)
    probes, maxabs = {}, {}
    rows = [r for r in FD.PROBE_ROWS if r < h.shape[1]]
    t0 = time.time()
    with torch.no_grad():
        for dl in range(hps.prior_depth):
            h = R.prior_layer(w, h, dl, hps, dtype=dtype)
            if dl + 1 in FD.PROBE_LAYERS:
                probes[dl + 1] = h[0, rows].numpy().copy()
                maxabs[dl + 1] = float(h.abs().max())
            if (dl + 1) % 6 == 0:
                print(f"  {tag} layer {dl + 1}/{hps.prior_depth}  {time.time() - t0:.0f}s  max|h| {float(h.abs().max()):.3f}", flush=True)
    return h[0], rows, probes, maxabs


def pooled(acts, latent_len, hps):
    acts = acts[:latent_len]
    return R.windowed_average(acts, frame_len(hps))[0], acts.mean(0)


def stage_case(name, w, hps):
    kind, idx, seconds, outlier = FD.WIDE_CASES[name]
    a = FD.jukebox_case_audio(kind, idx, seconds)
    latent_len = math.floor(hps.n_ctx * min(len(a), hps.sample_length) / hps.sample_length)
    ap = pad(a, hps)
    t0 = time.time()
    c, enc, _ = C.encode_codes(w, ap[None], hps, return_all=True)
    gap, _b, arg64 = gaps64(enc[0], w["bottleneck.level_blocks.2.k"])
    print(f"[{name}] encode {time.time() - t0:.1f}s, latent_audio_len {latent_len}, min gap {gap.min():.3e}, distinct codes {len(np.unique(c[0]))}", flush=True)
    acts, rows, probes, maxabs = run_layers(w, hps, c[0], torch.float32, tag=name)
    f10, f0 = pooled(acts.float(), latent_len, hps)
    OUT.update({
        f"{name}_codes": c[0].astype(np.int16), f"{name}_audio_sha": np.array(FD.sha(ap)), f"{name}_latent_len": np.int64(latent_len),
        f"{name}_min_gap": np.float64(gap.min()), f"{name}_agree_f64": np.float64((arg64 == c[0]).mean()),
        f"{name}_probe_rows": np.array(rows), f"{name}_probe_layers": np.array(sorted(probes)),
        f"{name}_probes": np.stack([probes[l] for l in sorted(probes)]), f"{name}_maxabs": np.array([maxabs[l] for l in sorted(probes)]),
        f"{name}_emb_f0": f0.numpy().astype(np.float32),
        f"{name}_acts_maxabs": np.float64(acts.abs().max()),
    })
    # round 6: every 4th pooled frame + the last one is kept (18 MB -> 7 MB in git; each frame is a mean over 34 tokens x all channels)
    keep = list(range(0, f10.shape[0], 4))
    if keep[-1] != f10.shape[0] - 1:
        keep.append(f10.shape[0] - 1)
    keep = np.array(keep, dtype=np.int64)
    OUT.update({f"{name}_emb_f10": f10.numpy().astype(np.float32)[keep], f"{name}_emb_f10_rows": keep, f"{name}_emb_f10_frames": np.int64(f10.shape[0]),
                f"{name}_emb_f10_maxabs": np.float64(f10.abs().max())})
    print(f"[{name}] emb_f10 {tuple(f10.shape)} max|emb| {float(f10.abs().max()):.3f} max|acts| {float(acts.abs().max()):.3f}", flush=True)
    return c[0], latent_len, f10


def stage_head64(name, w, hps, codes, latent_len, f10_fp32):
    """float64 over the first HEAD_TOKENS tokens; frames that lie wholly inside min(HEAD_TOKENS, latent_len)."""
    acts, rows, probes, _ = run_layers(w, hps, codes, torch.float64, tokens=FD.HEAD_TOKENS, tag=name + "/f64")
    n = min(FD.HEAD_TOKENS, latent_len)
    f10 = R.windowed_average(acts[:n], frame_len(hps))[0]
    keep = np.array(sorted({0, 7, 15, 22, f10.shape[0] - 1}), dtype=np.int64)       # five of the first frames (round 6: fixture size)
    OUT.update({f"head64_{name}_f10": f10.numpy().astype(np.float64)[keep], f"head64_{name}_f10_frames": keep, f"head64_{name}_rows": np.array(rows),
                f"head64_{name}_probes36": probes[36].astype(np.float64)})
    if f10_fp32 is not None:
        e = float((f10_fp32[: f10.shape[0]].double() - f10).abs().max())
        OUT[f"head64_{name}_fp32_oracle_err"] = np.float64(e)
        print(f"[{name}] fp32 oracle vs float64 on the first {f10.shape[0]} frames: max|err| {e:.3e}", flush=True)


def main():
    stages = sys.argv[1:] or ["codes", "rich", "short12", "base64", "outlier"]
    torch.set_num_threads(os.cpu_count() or 1)
    if os.path.exists(FD.WIDE_NPZ):
        OUT.update({k: v for k, v in np.load(FD.WIDE_NPZ).items()})
    hps = FD.jukebox_hps()
    t0 = time.time()
    w = FD.jukebox_weights_cpu(hps)
    xe_cal = C.encoder_forward(w, FD.jukebox_clip(FD.CAL_CLIP, hps)[None], hps)
    w["bottleneck.level_blocks.2.k"] = FD.codebook_from_encoding(xe_cal, hps)
    base = np.load(FD.JUKEBOX_NPZ)
    assert FD.sha(w["bottleneck.level_blocks.2.k"].numpy()) == str(base["codebook_sha"])
    OUT["codebook_sha"] = np.array(str(base["codebook_sha"]))
    print(f"weights + codebook: {time.time() - t0:.1f}s", flush=True)
    for st in stages:
        t0 = time.time()
        if st == "codes":
            stage_codes(w, hps)
        elif st == "base64":
            stage_head64("base", w, hps, base["codes"].astype(np.int64), hps.n_ctx, torch.from_numpy(base["emb_f10"]))
        elif st in ("rich", "short12"):
            codes, latent_len, f10 = stage_case(st, w, hps)
            save()
            stage_head64(st, w, hps, codes, latent_len, f10)
        elif st == "outlier":
            ch = FD.add_outlier_channels(w, hps)
            OUT["outlier_channels"] = np.array(ch)
            codes, latent_len, f10 = stage_case(st, w, hps)
            save()
            stage_head64(st, w, hps, codes, latent_len, f10)
        else:
            raise SystemExit(f"unknown stage {st}")
        print(f"stage {st}: {time.time() - t0:.0f}s", flush=True)
        save()


if __name__ == "__main__":
    main()
