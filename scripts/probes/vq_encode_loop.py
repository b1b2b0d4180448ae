#!/usr/bin/env python
"""Driver for the --pmc passes over the conv stack: N encode_top calls of the DEFAULT encoder (fused stages + near-tie certificate +
exact fix-up) on 8 synthetic 25 s clips, bracketed by two marker launches (llark_split16 on a 64-element tensor: `split16_v4_kernel`
appears nowhere else in the encoder) so that the counter rows of exactly those calls can be cut out by dispatch order.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o a -- python scripts/probes/vq_encode_loop.py [calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from llark_amd import ops
from llark_amd.jukebox.hparams import hparams_5b
from llark_amd.jukebox.synthetic import init_codebook_from_encodings, make_vqvae_weights, synthetic_clip
from llark_amd.jukebox.vqvae import VQVAE

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hps = hparams_5b()
w = make_vqvae_weights(hps, 0)


def clip(i):
    a = synthetic_clip(i, seconds=25.0)
    a = a / max(1e-12, float(np.abs(a).max()))
    a = a[: hps.sample_length]
    return np.pad(a, (0, max(0, hps.sample_length - len(a)))).astype(np.float32)


audio = torch.from_numpy(np.stack([clip(i) for i in range(8)])).cuda()
enc = VQVAE(hps, w, "cuda")
cal = VQVAE(hps, w, "cuda", exact=True).encoder_forward(torch.from_numpy(clip(100000)).cuda()[None, None, :])[0]
enc.set_codebook(init_codebook_from_encodings(cal.cpu(), hps.l_bins))
enc.encode_top(audio)
torch.cuda.synchronize()
marker = torch.ones(1, 64, device="cuda")
ops.split16(marker, torch.bfloat16, want_lo=False)
for _ in range(CALLS):
    enc.encode_top(audio)
ops.split16(marker, torch.bfloat16, want_lo=False)
torch.cuda.synchronize()
print("calls", CALLS, "near ties re-evaluated in the last call", enc.last_near_ties)
