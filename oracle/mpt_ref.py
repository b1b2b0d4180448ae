"""CPU ORACLE (test infrastructure, NOT product code) for the MPT backbone of the reference's ``m2t/models/mpt.py``
(SURVEY section 8(f) row 2): plain-torch fp32 restatement of

* ``WrappedMPTModel.forward`` (m2t/models/mpt.py:74-245): ``wte`` gather, ``mm_projector``, the audio splice (same rules and
  the same ValueErrors as the Llama wrapper -- shared with oracle/llama_ref.py:splice_audio);
* ``MPTModel.forward`` (m2t/llava/model/mpt/modeling_mpt.py:forward): no positional embedding when ALiBi is on, the
  ``(1, n_heads, 1, S)`` ALiBi bias of attention.py:462-483 sliced to the visible keys, the block stack, ``norm_f``;
* ``MPTBlock`` / ``MultiheadAttention`` / ``MPTMLP`` (blocks.py:60-86, attention.py:263-352,25-86): pre-LayerNorm, fused
  ``Wqkv`` (optional ``clip_qkv`` clamp, optional ``qk_ln`` LayerNorm over the full q / k vectors), causal softmax
  attention with the additive bias, ``out_proj``; LayerNorm, ``up_proj`` -> exact GELU -> ``down_proj``;
* ``MPTForCausalLM.forward`` (modeling_mpt.py:409-425): logits = hidden . wte^T (tied), optional ``logit_scale``,
  shifted cross entropy.

PINNING: the block arithmetic and the ALiBi bias are pinned against golden vectors produced by the REAL reference modules
(``MPTBlock`` and ``build_attn_bias`` imported from /root/reference/m2t/llava/model/mpt, tests/golden/make_mpt_golden.py ->
tests/golden/mpt_tiny.npz).  The reference's ``WrappedMPTForCausalLM`` itself cannot be constructed under the installed
transformers 5.15 (its PretrainedConfig subclass no longer stores its fields), so the model-level glue
above (embedding, splice, norm_f, tied logits, loss) is restated from the source, not executed: PARITY UNPINNED for that
glue only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .llama_ref import splice_audio


@dataclass
class MptSpec:
    d_model: int = 2048
    n_heads: int = 16
    n_layers: int = 24
    expansion_ratio: int = 4
    vocab_size: int = 50432
    max_seq_len: int = 2048
    alibi: bool = True
    alibi_bias_max: int = 8
    qk_ln: bool = False
    clip_qkv: Optional[float] = None
    no_bias: bool = True
    logit_scale: Optional[float] = None
    ln_eps: float = 1e-5
    mm_hidden_size: int = 512
    use_audio_start_end: bool = True
    audio_start_token: int = -1
    audio_end_token: int = -1
    audio_patch_token: int = -1


def alibi_slopes(n_heads: int, alibi_bias_max: int = 8) -> torch.Tensor:
    """attention.py:462-469."""
    n = 2 ** math.ceil(math.log2(n_heads))
    m = torch.arange(1, n + 1, dtype=torch.float32) * (alibi_bias_max / n)
    slopes = 1.0 / torch.pow(2, m)
    if n != n_heads:
        slopes = torch.cat([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes


def alibi_bias(n_heads: int, keys: int, alibi_bias_max: int = 8) -> torch.Tensor:
    """(1, n_heads, 1, keys): slope_h * (j - (keys - 1)) -- the last `keys` entries of the max_seq_len-long bias."""
    rel = torch.arange(1 - keys, 1, dtype=torch.float32).view(1, 1, 1, keys)
    return rel * alibi_slopes(n_heads, alibi_bias_max).view(1, n_heads, 1, 1)


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def block(w: Dict[str, torch.Tensor], spec: MptSpec, i: int, x: torch.Tensor, past=None):
    """x (B,S,D) -> (x', (k, v)); past = (k, v) each (B, S_past, D)."""
    p = f"transformer.blocks.{i}"
    D, H = spec.d_model, spec.n_heads
    hd = D // H
    g = lambda name: w.get(f"{p}.{name}")
    a = _ln(x, g("norm_1.weight"), g("norm_1.bias"), spec.ln_eps)
    qkv = F.linear(a, g("attn.Wqkv.weight"), g("attn.Wqkv.bias"))
    if spec.clip_qkv:
        qkv = qkv.clamp(min=-spec.clip_qkv, max=spec.clip_qkv)
    q, k, v = qkv.chunk(3, dim=2)
    if spec.qk_ln:
        q = _ln(q, g("attn.q_ln.weight"), g("attn.q_ln.bias"), spec.ln_eps)
        k = _ln(k, g("attn.k_ln.weight"), g("attn.k_ln.bias"), spec.ln_eps)
    if past is not None:
        k = torch.cat([past[0], k], dim=1)
        v = torch.cat([past[1], v], dim=1)
    B, S, _ = q.shape
    T = k.shape[1]
    qh = q.view(B, S, H, hd).transpose(1, 2)
    kh = k.view(B, T, H, hd).permute(0, 2, 3, 1)
    vh = v.view(B, T, H, hd).transpose(1, 2)
    att = qh.matmul(kh) * (1.0 / math.sqrt(hd))
    if spec.alibi:
        att = att + alibi_bias(H, T, spec.alibi_bias_max)
    causal = torch.ones(T, T, dtype=torch.bool).tril()[-S:, -T:]
    att = att.masked_fill(~causal.view(1, 1, S, T), torch.finfo(att.dtype).min)
    ctx = torch.softmax(att, dim=-1).matmul(vh).transpose(1, 2).reshape(B, S, D)
    x = x + F.linear(ctx, g("attn.out_proj.weight"), g("attn.out_proj.bias"))
    m = _ln(x, g("norm_2.weight"), g("norm_2.bias"), spec.ln_eps)
    up = F.gelu(F.linear(m, g("ffn.up_proj.weight"), g("ffn.up_proj.bias")))
    x = x + F.linear(up, g("ffn.down_proj.weight"), g("ffn.down_proj.bias"))
    return x, (k, v)


def forward(w: Dict[str, torch.Tensor], spec: MptSpec, input_ids: torch.Tensor, audio_encodings=None, labels=None,
            past_key_values=None, num_layers: Optional[int] = None, return_hidden: bool = False):
    """Returns dict(logits (B,S,V) fp32, loss or None, past_key_values [, hidden])."""
    if not spec.alibi:
        raise NotImplementedError("learned positions (wpe) are not restated: the reference's MPT configs use ALiBi")
    wte = w["transformer.wte.weight"]
    x = F.embedding(input_ids, wte)
    if audio_encodings is not None:
        proj = lambda a: F.linear(a, w["transformer.mm_projector.weight"], w["transformer.mm_projector.bias"])
        feats = [proj(a) for a in audio_encodings] if isinstance(audio_encodings, (list, tuple)) else proj(audio_encodings)
        x = splice_audio(input_ids, x, feats, spec)
    L = spec.n_layers if num_layers is None else num_layers
    new_past = []
    for i in range(L):
        x, kv = block(w, spec, i, x, None if past_key_values is None else past_key_values[i])
        new_past.append(kv)
    hidden = x
    x = _ln(x, w["transformer.norm_f.weight"], w.get("transformer.norm_f.bias"), spec.ln_eps)
    logits = F.linear(x, wte)
    if spec.logit_scale is not None:
        logits = logits * spec.logit_scale
    loss = None
    if labels is not None:
        shifted = torch.roll(labels, shifts=-1)                     # modeling_mpt.py:418-424
        shifted[:, -1] = -100
        loss = F.cross_entropy(logits.view(-1, logits.size(-1)), shifted.view(-1))
    out = dict(logits=logits, loss=loss, past_key_values=new_past)
    if return_hidden:
        out["hidden"] = hidden
    return out


def make_weights(spec: MptSpec, seed: int = 0, std: float = 0.05) -> Dict[str, torch.Tensor]:
    """Seeded bf16-VALUED fp32 weights under the reference's state-dict names."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g) * std).bfloat16().float()
    D, E = spec.d_model, spec.expansion_ratio * spec.d_model
    w = {"transformer.wte.weight": r(spec.vocab_size, D), "transformer.norm_f.weight": 1.0 + r(D),
         "transformer.mm_projector.weight": r(D, spec.mm_hidden_size), "transformer.mm_projector.bias": r(D)}
    for i in range(spec.n_layers):
        p = f"transformer.blocks.{i}"
        w[f"{p}.norm_1.weight"], w[f"{p}.norm_2.weight"] = 1.0 + r(D), 1.0 + r(D)
        w[f"{p}.attn.Wqkv.weight"], w[f"{p}.attn.out_proj.weight"] = r(3 * D, D), r(D, D)
        w[f"{p}.ffn.up_proj.weight"], w[f"{p}.ffn.down_proj.weight"] = r(E, D), r(D, E)
        if spec.qk_ln:
            w[f"{p}.attn.q_ln.weight"], w[f"{p}.attn.k_ln.weight"] = 1.0 + r(D), 1.0 + r(D)
        if not spec.no_bias:
            for name, n in (("norm_1", D), ("norm_2", D), ("attn.Wqkv", 3 * D), ("attn.out_proj", D), ("ffn.up_proj", E), ("ffn.down_proj", D)):
                w[f"{p}.{name}.bias"] = r(n)
            if spec.qk_ln:
                w[f"{p}.attn.q_ln.bias"], w[f"{p}.attn.k_ln.bias"] = r(D), r(D)
    if not spec.no_bias:
        w["transformer.norm_f.bias"] = r(D)
    return w


def greedy_generate(w, spec: MptSpec, input_ids, audio_encodings, max_new_tokens: int, eos_token_id=None):
    out = forward(w, spec, input_ids, audio_encodings)
    past = out["past_key_values"]
    ids = input_ids
    for _ in range(max_new_tokens):
        nxt = out["logits"][:, -1].argmax(-1, keepdim=True)
        ids = torch.cat([ids, nxt], dim=1)
        if eos_token_id is not None and bool((nxt == eos_token_id).all()):
            break
        out = forward(w, spec, nxt, None, past_key_values=past)
        past = out["past_key_values"]
    return ids
