#!/usr/bin/env python
"""Debug: where does the fused VQ-VAE encoder differ from the exact per-layer path?  (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from llark_amd import ops
from llark_amd.jukebox.hparams import hparams_5b, hparams_tiny
from llark_amd.jukebox.synthetic import init_codebook_from_encodings, make_vqvae_weights, synthetic_clip
from llark_amd.jukebox.vqvae import VQVAE
from oracle import jukebox_ref as R


def clip(hps, i, seconds):
    a = R.normalize_audio(synthetic_clip(i, seconds=seconds))[: hps.sample_length]
    return np.pad(a, (0, max(0, hps.sample_length - len(a)))).astype(np.float32)


for name, hps, secs in (("tiny", hparams_tiny(), 1.6), ("5b", hparams_5b(), 25.0)):
    w = make_vqvae_weights(hps, 0)
    vq = VQVAE(hps, w, "cuda")
    a = torch.from_numpy(np.stack([clip(hps, 0, secs)])).cuda()
    exact = vq.encoder_forward(a[:, None, :])
    vq.set_codebook(init_codebook_from_encodings(exact[0].cpu(), hps.l_bins))
    fused = vq.encoder_forward_fused(a)
    d = (fused - exact).abs()[0]                       # [64][T]
    scale = float(exact.abs().max())
    per_t = d.max(0).values
    top = per_t.topk(8)
    print(f"[{name}] T={d.shape[1]} max|x| {scale:.3f}  max diff {float(d.max()):.3e}  mean diff {float(d.mean()):.3e}  rms x {float(exact.pow(2).mean().sqrt()):.3f}")
    print("   worst positions:", [(int(i), f"{float(v):.2e}") for v, i in zip(top.values, top.indices)])
    print("   diff percentiles (per-position max): ", [f"{float(np.percentile(per_t.cpu().numpy(), q)):.2e}" for q in (50, 90, 99, 99.9, 100)])
    ce, de = ops.codebook_argmin(exact, vq.k, vq.kk, want_dist=True)
    cf, df = ops.codebook_argmin(fused, vq.k, vq.kk, want_dist=True)
    bad = (ce != cf)[0].nonzero().flatten().tolist()
    print(f"   code mismatches: {len(bad)} at {bad[:10]}")
    # per-stage check: run stages one by one against the exact path's intermediate activations
    taps = []
    vq.encoder_forward(a[:, None, :], taps=taps)
    # exact taps index: each stage = 1 conv + depth res (+1 out conv at the end of a level block)
    li, x, cin, t, slot = 0, a.contiguous(), 1, a.shape[1], 0
    for si, st in enumerate(vq.stages):
        c = 64 if st["wo_hi"] is not None else 32
        li += 1 + hps.depth + (1 if st["wo_hi"] is not None else 0)
        ref = taps[li - 1][0]                          # [c][t/2]
        out = torch.empty((1, c, t // 2), dtype=torch.float32, device="cuda")
        planes = vq._planes(slot, (t // 2) * c)
        ops.vqvae_stage(x, 1, cin, t, st, out_planes=planes, out_f32=out)
        dd = (out[0] - ref).abs()
        pt = dd.max(0).values
        tp = pt.topk(4)
        print(f"   stage {si}: cin {cin} T {t // 2} C {c}: max diff {float(dd.max()):.3e} (max|ref| {float(ref.abs().max()):.3f}) worst at {[int(i) for i in tp.indices]}; "
              f"planes vs f32: {float(((planes[0][: (t // 2) * c].float() + planes[1][: (t // 2) * c].float()).view(t // 2, c).t() - out[0]).abs().max()):.2e}")
        x, cin, t, slot = planes, c, t // 2, slot ^ 1
