#!/usr/bin/env python
"""Generates tests/golden/mpt_tiny.npz by running the REAL reference modules (this container only; /root/reference does
not exist on the GPU box): ``MPTBlock`` (m2t/llava/model/mpt/blocks.py) and ``build_attn_bias`` / ``attn_bias_shape``
(attention.py), chained exactly as ``MPTModel.forward`` chains them (modeling_mpt.py: attn_bias built once for
max_seq_len, each block called with (x, past_key_value, attn_bias, attention_mask=None, is_causal=True)).

Two cases: A = ALiBi only (the shipped MPT configs), B = ALiBi + qk_ln + clip_qkv + biases.  For each: prefill over S
tokens, then one cached decode step.  Usage:  PYTHONPATH=/root/reference python tests/golden/make_mpt_golden.py
"""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")

from m2t.llava.model.mpt.attention import attn_bias_shape, build_attn_bias  # noqa: E402
from m2t.llava.model.mpt.blocks import MPTBlock  # noqa: E402

from oracle import mpt_ref as MR  # noqa: E402  (only to draw the seeded weights under the reference's names)


def run_case(tag, spec, out):
    w = MR.make_weights(spec, seed=3 if tag == "A" else 4)
    attn_config = dict(attn_type="multihead_attention", attn_pdrop=0.0, attn_impl="torch", qk_ln=spec.qk_ln, clip_qkv=spec.clip_qkv,
                       softmax_scale=None, prefix_lm=False, attn_uses_sequence_id=False, alibi=True, alibi_bias_max=spec.alibi_bias_max)
    blocks = []
    for i in range(spec.n_layers):
        blk = MPTBlock(d_model=spec.d_model, n_heads=spec.n_heads, expansion_ratio=spec.expansion_ratio, attn_config=attn_config,
                       norm_type="layernorm")             # fp32 CPU path: LPLayerNorm == LayerNorm without autocast
        sd = {k[len(f"transformer.blocks.{i}."):]: v for k, v in w.items() if k.startswith(f"transformer.blocks.{i}.")}
        if spec.no_bias:                                   # MPTModel.__init__ drops every bias for no_bias models
            for m in blk.modules():
                if hasattr(m, "bias") and isinstance(m.bias, torch.nn.Parameter):
                    m.register_parameter("bias", None)
        missing, unexpected = blk.load_state_dict(sd, strict=True), None
        blocks.append(blk.eval())
    shape = attn_bias_shape("torch", spec.n_heads, spec.max_seq_len, True, prefix_lm=False, causal=True, use_sequence_id=False)
    bias = build_attn_bias("torch", torch.zeros(shape), spec.n_heads, spec.max_seq_len, causal=True, alibi=True,
                           alibi_bias_max=spec.alibi_bias_max)
    g = torch.Generator().manual_seed(11)
    B, S = 2, 37
    x0 = torch.randn(B, S, spec.d_model, generator=g)
    x1 = torch.randn(B, 1, spec.d_model, generator=g)
    with torch.no_grad():
        x, pasts = x0, []
        for blk in blocks:
            x, pkv = blk(x, past_key_value=(), attn_bias=bias, attention_mask=None, is_causal=True)
            pasts.append(pkv)
        y0 = x
        x = x1
        for blk, pkv in zip(blocks, pasts):
            x, _ = blk(x, past_key_value=pkv, attn_bias=bias, attention_mask=None, is_causal=True)
        y1 = x
    out[f"{tag}_x0"], out[f"{tag}_x1"], out[f"{tag}_y0"], out[f"{tag}_y1"] = x0.numpy(), x1.numpy(), y0.numpy(), y1.numpy()
    out[f"{tag}_bias_last64"] = bias[..., -64:].numpy()
    # weights are NOT stored: oracle.mpt_ref.make_weights(spec, seed) reproduces them bit for bit (CPU torch.Generator);
    # a checksum guards against a silent change of that generator
    out[f"{tag}_wsum"] = np.array([float(sum(v.double().sum() for v in w.values()))])


def main():
    out = {}
    base = dict(d_model=256, n_heads=2, n_layers=2, expansion_ratio=4, vocab_size=96, max_seq_len=128, mm_hidden_size=64)
    run_case("A", MR.MptSpec(**base), out)
    run_case("B", MR.MptSpec(**base, qk_ln=True, clip_qkv=0.75, no_bias=False, alibi_bias_max=4), out)
    # a 3-head slope vector exercises the non-power-of-two branch of gen_slopes
    from m2t.llava.model.mpt.attention import gen_slopes
    out["slopes_3"] = gen_slopes(3, 8).reshape(-1).numpy()
    out["slopes_16"] = gen_slopes(16, 8).reshape(-1).numpy()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpt_tiny.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
