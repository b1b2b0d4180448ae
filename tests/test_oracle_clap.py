"""oracle/clap_ref.py against transformers.ClapAudioModelWithProjection (independent public port of laion's HTSAT audio
branch) on golden vectors: tests/golden/make_clap_golden.py -> clap_tiny.npz."""
import os

import numpy as np
import torch

from oracle import clap_ref as CR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clap_tiny.npz")
SPEC = CR.ClapSpec(embed_dim=32, depths=[2, 2, 2, 1], heads=[1, 2, 4, 8], proj_dim=64)


def test_htsat_matches_independent_port():
    z = np.load(GOLD)
    w = CR.make_weights(SPEC, seed=5)
    assert abs(float(sum(v.double().sum() for v in w.values())) - float(z["wsum"][0])) < 1e-5, "weight generator drifted"
    x = torch.from_numpy(z["x"])
    y = CR.forward(w, SPEC, x, normalize=False)
    ref = torch.from_numpy(z["audio_embeds"])
    assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), (y - ref).abs().max().item()
    yn = CR.forward(w, SPEC, x)                                          # laion's get_audio_embedding normalises
    assert torch.allclose(yn.norm(dim=-1), torch.ones(2), atol=1e-6)
    assert torch.allclose(yn, torch.nn.functional.normalize(ref, dim=-1), atol=2e-5)


def test_structure_properties():
    """Relative-position index is the Swin one (symmetric offsets share a slot), the image fold keeps every sample, odd
    blocks are the shifted ones (a constant input stays translation-consistent), base spec widths."""
    idx = CR.relative_position_index(8)
    assert idx.shape == (64, 64) and int(idx.min()) == 0 and int(idx.max()) == 15 * 15 - 1 and int(idx[0, 0]) == int(idx[5, 5]) == 7 * 15 + 7
    spec = CR.ClapSpec()
    assert spec.freq_ratio == 4 and spec.out_width == 1024
    x = torch.arange(2 * 1024 * 64, dtype=torch.float32).view(2, 1, 1024, 64)
    img = CR.mel_to_image(x, spec)
    assert img.shape == (2, 1, 256, 256) and torch.equal(img.flatten(1).sort().values, x.flatten(1).sort().values)
    assert torch.equal(img[0, 0, :64, :], x[0, 0, :256, :].t()) and torch.equal(img[0, 0, 64:128, :], x[0, 0, 256:512, :].t())
