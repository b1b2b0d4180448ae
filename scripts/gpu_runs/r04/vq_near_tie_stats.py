#!/usr/bin/env python
"""Round 4: what the near-tie certificate of the fused VQ-VAE encoder has to cover, measured (10 clips: bench.py's 8 + 2).

Per token: e = x_fused - x_exact (encoder output, 64 channels), |e|_2 / |x|_2; the best / second-best gap of the exact
evaluation and of the fused one; how far the gap of that code pair moved; which tokens flip.  Then, for a grid of
(TIE_E_REL, TIE_ULPS), how many tokens the certificate flags, whether every flip is among them, and what the default
path costs with the fix-up inside the timed region (HIP events, medians) next to the plain fused argmin and the exact path.
Run on the GPU box:  python scripts/gpu_runs/r04/vq_near_tie_stats.py"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import fulldepth as FD
from llark_amd import ops
from llark_amd.jukebox.synthetic import make_vqvae_weights
from llark_amd.jukebox.vqvae import VQVAE

hps = FD.jukebox_hps()
w = make_vqvae_weights(hps, 0)
exact = VQVAE(hps, w, "cuda", exact=True)
k = FD.codebook_from_encoding(exact.encoder_forward(torch.from_numpy(FD.jukebox_clip(FD.CAL_CLIP, hps)).cuda()[None, None, :])[0].cpu(), hps)
exact.set_codebook(k)
raw = VQVAE(hps, w, "cuda", tie_e_rel=None)
raw.set_codebook(k)
clips = list(range(10))
audio = torch.from_numpy(np.stack([FD.jukebox_clip(i, hps) for i in clips])).cuda()
k64 = exact.k.double()
kk64 = (k64 ** 2).sum(1)

stats = {}
rel_e, dgap, gap_e, gap_f, flips = [], [], [], [], []
for i in clips:
    a = audio[i:i + 1]
    xe = exact.encoder_forward(a[:, None, :])[0]                    # (64, 8192) exact fp32
    xf = raw.encoder_forward_fused(a)[0]
    ce, de = ops.codebook_argmin(xe[None], exact.k, exact.kk, want_dist=True)
    cf, _ = ops.codebook_argmin(xf[None].contiguous(), exact.k, exact.kk, want_dist=True)
    e = (xf - xe).double()
    xn = xe.double().pow(2).sum(0).sqrt()
    rel_e.append((e.pow(2).sum(0).sqrt() / xn).cpu())
    # float64 distances of both encodings (real-arithmetic stand-in)
    def dist(x):
        x = x.double().t()
        return (x ** 2).sum(1, keepdim=True) - 2 * x @ k64.t() + kk64[None]
    d_e, d_f = dist(xe), dist(xf)
    top = torch.topk(d_e, 2, dim=1, largest=False)
    a_idx, b_idx = top.indices[:, 0], top.indices[:, 1]
    g_e = top.values[:, 1] - top.values[:, 0]
    g_f_pair = d_f.gather(1, b_idx[:, None])[:, 0] - d_f.gather(1, a_idx[:, None])[:, 0]
    dgap.append((g_f_pair - g_e).cpu())                           # movement of THAT pair's gap caused by e (float64: no distance rounding)
    gap_e.append(g_e.cpu())
    topf = torch.topk(d_f, 2, dim=1, largest=False)
    gap_f.append((topf.values[:, 1] - topf.values[:, 0]).cpu())
    fl = (ce[0] != cf[0]).nonzero()[:, 0].cpu()
    flips.append(fl)
    # the fp32 rounding of the distance chain itself: exact kernel's fp32 distance vs float64, on the winning code
    rnd = (de[0].double() - d_e.gather(1, ce[0][:, None])[:, 0]).abs()
    print(f"clip {i}: |e|/|x| max {float(rel_e[-1].max()):.3e} median {float(rel_e[-1].median()):.3e}; max|e| {float(e.abs().max()):.3e}; "
          f"|x| rms {float(xn.pow(2).mean().sqrt()):.2f}; pair-gap movement by e: max {float(dgap[-1].abs().max()):.3e} rms {float(dgap[-1].pow(2).mean().sqrt()):.3e}; "
          f"fp32 distance rounding (winning code) max {float(rnd.max()):.3e} rms {float(rnd.pow(2).mean().sqrt()):.3e}; "
          f"flips {len(fl)} at exact gaps {[f'{float(g_e[t]):.2e}' for t in fl]}", flush=True)
rel_e, dgap, gap_e, gap_f = torch.cat(rel_e), torch.cat(dgap), torch.cat(gap_e), torch.cat(gap_f)
print(f"ALL 10 clips ({rel_e.numel()} tokens): |e|/|x| max {float(rel_e.max()):.3e}, 99.9 % {float(rel_e.quantile(0.999)):.3e}, median {float(rel_e.median()):.3e}; "
      f"pair-gap movement max {float(dgap.abs().max()):.3e}, rms {float(dgap.pow(2).mean().sqrt()):.3e}; total flips {sum(len(f) for f in flips)}")
for b in (1e-3, 3e-3, 1e-2, 3e-2, 1e-1):
    print(f"  tokens with exact gap < {b:.0e}: {int((gap_e < b).sum())}; fused gap < {b:.0e}: {int((gap_f < b).sum())}")

# certificate grid: flagged counts on bench clips 0..7 (one batch of 8), and mismatches vs the exact path
a8 = audio[:8].contiguous()
ce8 = exact.encode_top(a8)
print("\ncertificate grid on the bench batch (8 clips, 65536 tokens):")
grid = {}
for e_rel in (0.5e-6, 1e-6, 2e-6, 4e-6):
    for ulps in (4.0, 16.0, 64.0):
        vq = VQVAE(hps, w, "cuda", tie_e_rel=e_rel, tie_ulps=ulps)
        vq.set_codebook(k)
        c = vq.encode_top(a8)
        mism = int((c != ce8).sum())
        grid[f"{e_rel:g}/{ulps:g}"] = (vq.last_near_ties, mism)
        print(f"  TIE_E_REL {e_rel:g} TIE_ULPS {ulps:g}: flagged {vq.last_near_ties}, mismatches vs exact {mism}", flush=True)

# timing
def timeit(m, rounds=7, reps=3):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            m.encode_top(a8)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)

gb = raw.algorithmic_bytes(8) / 1e9
res = {"grid": grid}
print("\ntiming, 8 clips (median of 7 x 3 calls):")
for name, m in (("fused argmin, no certificate", raw), ("default (TIE_E_REL 2e-6, 16 ulps)", VQVAE(hps, w, "cuda")),
                ("wide (4e-6, 64 ulps)", VQVAE(hps, w, "cuda", tie_e_rel=4e-6, tie_ulps=64.0)),
                ("narrow (1e-6, 4 ulps)", VQVAE(hps, w, "cuda", tie_e_rel=1e-6, tie_ulps=4.0)), ("exact", exact)):
    m.set_codebook(k)
    m.encode_top(a8)
    ms = timeit(m)
    res[name] = {"ms": round(ms, 3), "frac_of_8tbs": round(gb / ms * 1e3 / 8000, 4), "near_ties": getattr(m, "last_near_ties", 0)}
    print(f"  {name}: {ms:.3f} ms = {gb / ms * 1e3 / 8000:.3f} of 8 TB/s algorithmic; near-ties {getattr(m, 'last_near_ties', 0)}", flush=True)
print(json.dumps(res))
