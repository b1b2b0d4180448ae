#!/usr/bin/env python
"""Generates tests/golden/clap_tiny.npz with transformers.ClapAudioModelWithProjection (an independent public port of laion's
HTSAT audio branch; this container only): small configuration, weights drawn by oracle.clap_ref.make_weights (not stored:
reproducible bit for bit, guarded by a checksum) and loaded into the HF model with strict=True.
Usage: python tests/golden/make_clap_golden.py"""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
from transformers import ClapAudioConfig, ClapAudioModelWithProjection  # noqa: E402

from oracle import clap_ref as CR  # noqa: E402

SPEC = CR.ClapSpec(embed_dim=32, depths=[2, 2, 2, 1], heads=[1, 2, 4, 8], proj_dim=64)


def main():
    cfg = ClapAudioConfig(patch_embeds_hidden_size=SPEC.embed_dim, depths=SPEC.depths, num_attention_heads=SPEC.heads,
                          hidden_size=SPEC.out_width, projection_dim=SPEC.proj_dim, window_size=SPEC.window, spec_size=SPEC.spec_size,
                          num_mel_bins=SPEC.mel_bins, enable_fusion=False, layer_norm_eps=SPEC.ln_eps, projection_hidden_act="relu")
    m = ClapAudioModelWithProjection(cfg).eval()
    w = CR.make_weights(SPEC, seed=5)
    sd = dict(w)
    for k, v in m.state_dict().items():                                   # buffers that are functions of the config
        if k.endswith("relative_position_index") or k.endswith("num_batches_tracked"):
            sd[k] = v
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 1, 1001, SPEC.mel_bins, generator=g) * 20 - 30          # log-mel-like range (dB)
    with torch.no_grad():
        out = m(input_features=x)
        pooled = m.audio_model(input_features=x).pooler_output
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clap_tiny.npz")
    np.savez_compressed(dst, x=x.numpy().astype(np.float32), audio_embeds=out.audio_embeds.numpy(), pooled=pooled.numpy(),
                        wsum=np.array([float(sum(v.double().sum() for v in w.values()))]))
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
