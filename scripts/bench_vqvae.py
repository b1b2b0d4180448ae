#!/usr/bin/env python
"""Times the VQ-VAE level-2 encoder + codebook (8 synthetic 25 s clips resident in HBM): fused stage kernels (default) vs the exact
per-layer fp32 path, HIP events around whole encode_top calls, interleaved rounds; also counts code mismatches between the two.
Run on the GPU box:  python scripts/bench_vqvae.py [clips]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from llark_amd.jukebox.hparams import hparams_5b
from llark_amd.jukebox.synthetic import init_codebook_from_encodings, make_vqvae_weights, synthetic_clip
from llark_amd.jukebox.vqvae import VQVAE
from oracle import jukebox_ref as R

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hps = hparams_5b()
w = make_vqvae_weights(hps, 0)


def clip(i):
    a = R.normalize_audio(synthetic_clip(i, seconds=25.0))[: hps.sample_length]
    return np.pad(a, (0, max(0, hps.sample_length - len(a)))).astype(np.float32)


audio = torch.from_numpy(np.stack([clip(i) for i in range(N)])).cuda()
fused, exact = VQVAE(hps, w, "cuda"), VQVAE(hps, w, "cuda", exact=True)
cal = exact.encoder_forward(torch.from_numpy(clip(100000)).cuda()[None, None, :])[0]
k = init_codebook_from_encodings(cal.cpu(), hps.l_bins)
fused.set_codebook(k)
exact.set_codebook(k)
cf, ce = fused.encode_top(audio), exact.encode_top(audio)
torch.cuda.synchronize()
mism = int((cf != ce).sum())
times = {"fused": [], "exact": []}
for _ in range(7):
    for name, m in (("fused", fused), ("exact", exact)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            m.encode_top(audio)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
gb = fused.algorithmic_bytes(N) / 1e9
res = {"clips": N, "code_mismatches_fused_vs_exact": mism, "tokens": int(cf.numel()), "algorithmic_gb": round(gb, 3)}
for name in times:
    med = statistics.median(times[name])
    res[name] = {"ms": round(med, 3), "ms_per_clip": round(med / N, 4), "algorithmic_gbs": round(gb / med * 1e3, 1), "frac_of_8tbs": round(gb / med * 1e3 / 8000, 4)}
    print(f"{name}: {med:.3f} ms for {N} clips = {med / N * 1e3:.1f} us/clip; {gb / med * 1e3:.0f} GB/s algorithmic = {gb / med * 1e3 / 8000:.3f} of 8 TB/s", flush=True)
print(f"code mismatches fused vs exact: {mism} of {cf.numel()}")
print(json.dumps(res))
