"""MPT backbone on the HIP kernels vs the oracle (itself pinned against the real reference MPTBlock / ALiBi bias):
SURVEY section 8(f) row 2."""
import dataclasses

import numpy as np
import pytest
import torch

from conftest import report_close

pytestmark = pytest.mark.gpu

BASE = dict(d_model=256, n_heads=2, n_layers=2, expansion_ratio=4, vocab_size=96, max_seq_len=128, mm_hidden_size=64)


def _engine(spec, w, precision, max_batch=2, max_seq=64):
    from llark_amd.m2t.mpt_engine import HipMptEngine, MptDims
    dims = MptDims(d_model=spec.d_model, n_heads=spec.n_heads, n_layers=spec.n_layers, expansion_ratio=spec.expansion_ratio,
                   vocab_size=spec.vocab_size, max_seq_len=spec.max_seq_len, alibi_bias_max=spec.alibi_bias_max, qk_ln=spec.qk_ln,
                   clip_qkv=spec.clip_qkv, logit_scale=spec.logit_scale, ln_eps=spec.ln_eps, mm_hidden_size=spec.mm_hidden_size)
    eng = HipMptEngine(dims, "cuda", max_batch, max_seq, precision=precision)
    eng.load_state_dict(w)
    return eng


@pytest.mark.parametrize("case", ["alibi", "alibi+qk_ln+clip+bias+logit_scale"])
def test_mpt_engine_split_matches_oracle(case):
    """fp32-class mode: logits, hidden state and cached decode steps vs the fp32 oracle (tolerance 1e-4 of max|ref|;
    measured ~2e-5), with the audio splice through mm_projector."""
    from oracle import mpt_ref as MR
    extra = {} if case == "alibi" else dict(qk_ln=True, clip_qkv=0.75, no_bias=False, alibi_bias_max=4, logit_scale=0.5)
    spec = MR.MptSpec(**BASE, audio_start_token=93, audio_end_token=94, audio_patch_token=95, **extra)
    w = MR.make_weights(spec, seed=7)
    g = torch.Generator().manual_seed(2)
    B, S = 2, 29
    ids = torch.randint(0, 90, (B, S), generator=g)
    ids[:, 2], ids[:, 3:8], ids[:, 8] = 93, 95, 94
    aud = torch.randn(B, 5, 64, generator=g)
    ref = MR.forward(w, spec, ids, aud, return_hidden=True)
    eng = _engine(spec, w, "split")
    segs = [(b, 2, aud[b].cuda()) for b in range(B)]
    logits = eng.forward_tokens(ids.cuda(), segs)
    report_close("mpt logits (split) vs oracle", logits.cpu(), ref["logits"], 1e-4 * ref["logits"].abs().max().item())
    # cached decode: 4 steps, each compared with the oracle run on the growing cache
    past = ref["past_key_values"]
    for step in range(4):
        tok = torch.randint(0, 90, (B, 1), generator=g)
        r = MR.forward(w, spec, tok, None, past_key_values=past)
        past = r["past_key_values"]
        lg = eng.forward_tokens(tok.cuda(), (), pos0=eng.cur_len)
        report_close(f"mpt decode step {step}", lg.cpu(), r["logits"], 1e-4 * r["logits"].abs().max().item())
    assert eng.cur_len == S + 4
    # last_only prefill == last row of the full prefill
    lo = eng.forward_tokens(ids.cuda(), segs, last_only=True)
    report_close("last_only", lo[:, 0].cpu(), ref["logits"][:, -1], 1e-4 * ref["logits"].abs().max().item())


def test_mpt_engine_bf16_flow_and_errors():
    """Single-pass mode stays within bf16-flow distance of the fp32 oracle; sequence longer than the cache raises."""
    from oracle import mpt_ref as MR
    spec = MR.MptSpec(**BASE)
    w = MR.make_weights(spec, seed=9)
    ids = torch.randint(0, 96, (2, 40), generator=torch.Generator().manual_seed(4))
    ref = MR.forward(w, spec, ids)["logits"]
    eng = _engine(spec, w, "bf16")
    lg = eng.forward_tokens(ids.cuda())
    rel = (lg.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 3e-2, rel
    assert (lg.cpu().argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9
    with pytest.raises(ValueError, match="Cannot forward input"):
        eng.forward_tokens(torch.zeros((2, 65), dtype=torch.int64, device="cuda"))
    with pytest.raises(NotImplementedError):
        from llark_amd.m2t.mpt_engine import HipMptEngine, MptDims
        HipMptEngine(MptDims(d_model=256, n_heads=4), "cuda")               # head_dim 64


def test_mpt_1b_width_two_blocks():
    """MPT-1B widths (d_model 2048, 16 heads, expansion 4, ALiBi), 2 blocks, S = 371-like prompt, small vocab."""
    from oracle import mpt_ref as MR
    spec = MR.MptSpec(d_model=2048, n_heads=16, n_layers=2, expansion_ratio=4, vocab_size=512, max_seq_len=2048, mm_hidden_size=512,
                      audio_start_token=509, audio_end_token=510, audio_patch_token=511)
    w = MR.make_weights(spec, seed=1, std=0.02)
    g = torch.Generator().manual_seed(0)
    B, S = 2, 131
    ids = torch.randint(0, 500, (B, S), generator=g)
    ids[:, 1], ids[:, 2], ids[:, 3] = 509, 511, 510                         # CLAP-style: ONE 512-d frame per clip (F = 1)
    aud = torch.randn(B, 1, 512, generator=g)
    ref = MR.forward(w, spec, ids, aud)["logits"]
    eng = _engine(spec, w, "split", max_batch=B, max_seq=192)
    lg = eng.forward_tokens(ids.cuda(), [(b, 1, aud[b].cuda()) for b in range(B)])
    report_close("mpt-1b width logits vs oracle", lg.cpu(), ref, 1e-4 * ref.abs().max().item())
