"""oracle/mpt_ref.py against golden vectors produced by the REAL reference MPT modules (tests/golden/make_mpt_golden.py:
MPTBlock + build_attn_bias from /root/reference/m2t/llava/model/mpt), plus the restated model-level glue."""
import os

import numpy as np
import pytest
import torch

from oracle import mpt_ref as MR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpt_tiny.npz")
BASE = dict(d_model=256, n_heads=2, n_layers=2, expansion_ratio=4, vocab_size=96, max_seq_len=128, mm_hidden_size=64)
CASES = {"A": (MR.MptSpec(**BASE), 3), "B": (MR.MptSpec(**BASE, qk_ln=True, clip_qkv=0.75, no_bias=False, alibi_bias_max=4), 4)}


@pytest.mark.parametrize("tag", ["A", "B"])
def test_block_stack_matches_reference_modules(tag):
    z = np.load(GOLD)
    spec, seed = CASES[tag]
    w = MR.make_weights(spec, seed=seed)
    assert abs(float(sum(v.double().sum() for v in w.values())) - float(z[f"{tag}_wsum"][0])) < 1e-6, "weight generator drifted"
    x = torch.from_numpy(z[f"{tag}_x0"])
    pasts = []
    for i in range(spec.n_layers):
        x, kv = MR.block(w, spec, i, x, None)
        pasts.append(kv)
    ref0 = torch.from_numpy(z[f"{tag}_y0"])
    assert (x - ref0).abs().max().item() <= 2e-5 * ref0.abs().max().item()
    x = torch.from_numpy(z[f"{tag}_x1"])                              # one cached decode step
    for i in range(spec.n_layers):
        x, _ = MR.block(w, spec, i, x, pasts[i])
    ref1 = torch.from_numpy(z[f"{tag}_y1"])
    assert (x - ref1).abs().max().item() <= 2e-5 * ref1.abs().max().item()
    # the ALiBi bias itself: last 64 key positions of the reference's max_seq_len-long tensor
    got = MR.alibi_bias(spec.n_heads, 64, spec.alibi_bias_max)
    assert torch.allclose(got, torch.from_numpy(z[f"{tag}_bias_last64"]).float(), rtol=0, atol=1e-6)


def test_alibi_slopes_match_gen_slopes():
    z = np.load(GOLD)
    assert torch.equal(MR.alibi_slopes(3, 8), torch.from_numpy(z["slopes_3"]))       # non-power-of-two interleave
    assert torch.equal(MR.alibi_slopes(16, 8), torch.from_numpy(z["slopes_16"]))


def test_model_level_glue_properties():
    """Restated glue (unpinned): tied logits, logit_scale, shifted loss, cache == no cache, splice errors."""
    spec = MR.MptSpec(**BASE, audio_start_token=93, audio_end_token=94, audio_patch_token=95)
    w = MR.make_weights(spec, seed=5)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 90, (2, 21), generator=g)
    ids[:, 3], ids[:, 4:9], ids[:, 9] = 93, 95, 94
    aud = torch.randn(2, 5, 64, generator=g)
    out = MR.forward(w, spec, ids, aud, labels=ids.clone())
    assert out["logits"].shape == (2, 21, 96) and torch.isfinite(out["loss"])
    # audio rows replaced the patch embeddings: changing a patch-token id must not change anything
    ids2 = ids.clone()
    ids2[:, 5] = 7
    with pytest.raises(ValueError):                                   # ... but it breaks the start/end adjacency rule? no: count rule
        bad = ids.clone()
        bad[0, 9] = 5                                                 # end token missing
        MR.forward(w, spec, bad, aud)
    # prefill + cached steps == one full forward
    full = MR.forward(w, spec, ids, aud)["logits"]
    part = MR.forward(w, spec, ids[:, :15], aud)
    step = MR.forward(w, spec, ids[:, 15:], None, past_key_values=part["past_key_values"])
    assert torch.allclose(torch.cat([part["logits"], step["logits"]], 1), full, atol=2e-5 * full.abs().max().item())
    # loss = CE(logits[:, :-1], labels[:, 1:])
    lab = ids.clone()
    lab[:, :10] = -100
    o = MR.forward(w, spec, ids, aud, labels=lab)
    ref = torch.nn.functional.cross_entropy(o["logits"][:, :-1].reshape(-1, 96), lab[:, 1:].reshape(-1), ignore_index=-100)
    assert abs(o["loss"].item() - ref.item()) < 1e-6
    import dataclasses
    o2 = MR.forward(w, dataclasses.replace(spec, logit_scale=0.5), ids, aud)
    assert torch.allclose(o2["logits"], 0.5 * full)
