"""CPU ORACLE (test infrastructure, NOT product code) for the LLM half of LLark's hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Restates, in plain torch on CPU:
  * the audio splice of ``WrappedLlamav2Model.forward``      (m2t/models/llamav2.py:95-234),
  * ``mm_projector`` = nn.Linear(mm_hidden_size, hidden_size) (m2t/models/llamav2.py:79,133),
  * ``WrappedLlamav2ForCausalLM.forward``: lm_head + shifted CE loss (m2t/models/llamav2.py:259-337),
  * the Llama-2 decoder arithmetic the wrapper delegates to HF ``LlamaModel.forward``
    (transformers==4.29.2 pinned in requirements.txt:4; RMSNorm, half-split RoPE, eager attention
    with fp32 softmax, SwiGLU MLP -- SURVEY Appendix B).

PINNED: tests/golden/llama_*.npz were generated in the build container by importing the REAL
reference wrapper (``/root/reference/m2t/models/llamav2.py`` over the installed transformers,
fp32, eager attention) with tests/golden/make_llama_golden.py; tests/test_oracle_llama.py checks this
restatement against them (<= 2e-5 abs on logits, exact on errors raised).

``act_dtype=torch.bfloat16`` mirrors the reduced-precision points of the reference's GPU dtype flow
(bf16 autocast / bf16 weights: inputs of every Linear, q/k/v after RoPE, and the softmax
probabilities are rounded to bf16 -- each of these is also a rounding point in HF's bf16 run) while
all accumulation stays fp32.  The HIP path computes in this flow except that it keeps the softmax
probabilities at >= 16 bits (``round_probs=False``).  ``act_dtype=None`` is the pure-fp32 CPU path
of the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn.functional as F


@dataclass
class LlamaSpec:
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    vocab_size: int = 32000
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    mm_hidden_size: int = 4800
    audio_start_token: int = -1
    audio_end_token: int = -1
    audio_patch_token: int = -1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def _rnd(x: torch.Tensor, act_dtype) -> torch.Tensor:
    return x if act_dtype is None else x.to(act_dtype).to(torch.float32)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """LlamaRMSNorm: x * rsqrt(mean(x^2) + eps) * weight, in fp32."""
    x = x.float()
    var = x.pow(2).mean(-1, keepdim=True)
    return w.float() * (x * torch.rsqrt(var + eps))


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float):
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = positions.float()[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def splice_audio(input_ids: torch.Tensor, inputs_embeds: torch.Tensor, audio_features, spec: LlamaSpec,
                 has_past: bool = False):
    """m2t/models/llamav2.py:141-222 (use_audio_start_end=True, orig_embeds_params=None branch)."""
    new_input_embeds = []
    cur_audio_idx = 0
    for cur_input_ids, cur_input_embeds in zip(input_ids, inputs_embeds):
        if (cur_input_ids == spec.audio_start_token).sum() != (cur_input_ids == spec.audio_end_token).sum():
            raise ValueError("The number of image start tokens and image end tokens should be the same.")
        audio_start_tokens = torch.where(cur_input_ids == spec.audio_start_token)[0]
        if len(audio_start_tokens):
            cur_new = cur_input_embeds
            for pos in audio_start_tokens:
                cur_audio_features = audio_features[cur_audio_idx]
                num_frames = cur_audio_features.shape[0]
                if cur_input_ids[pos + num_frames + 1] != spec.audio_end_token:
                    raise ValueError("The image end token should follow the image start token.")
                cur_new = torch.cat((cur_input_embeds[: pos + 1], cur_audio_features,
                                     cur_input_embeds[pos + num_frames + 1:]), dim=0)
                cur_audio_idx += 1
            new_input_embeds.append(cur_new)
        else:
            new_input_embeds.append(cur_input_embeds)
    return torch.stack(new_input_embeds, dim=0)


def decoder_layer(h: torch.Tensor, w: Dict[str, torch.Tensor], i: int, spec: LlamaSpec, cos, sin, mask, act_dtype,
                  past_kv=None, round_probs: bool = True):
    """One LlamaDecoderLayer (pre-norm residual x2).  h: (B,S,H) fp32."""
    B, S, H = h.shape
    nh, hd = spec.num_attention_heads, spec.head_dim
    p = f"model.layers.{i}"
    x = _rnd(rmsnorm(h, w[f"{p}.input_layernorm.weight"], spec.rms_norm_eps), act_dtype)
    q = F.linear(x, w[f"{p}.self_attn.q_proj.weight"].float()).view(B, S, nh, hd).transpose(1, 2)
    k = F.linear(x, w[f"{p}.self_attn.k_proj.weight"].float()).view(B, S, nh, hd).transpose(1, 2)
    v = F.linear(x, w[f"{p}.self_attn.v_proj.weight"].float()).view(B, S, nh, hd).transpose(1, 2)
    q = q * cos + _rotate_half(q) * sin
    k = k * cos + _rotate_half(k) * sin
    q, k, v = _rnd(q, act_dtype), _rnd(k, act_dtype), _rnd(v, act_dtype)
    if past_kv is not None:
        k = torch.cat((past_kv[0], k), dim=2)
        v = torch.cat((past_kv[1], v), dim=2)
    att = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    att = att + mask
    att = F.softmax(att, dim=-1, dtype=torch.float32)
    if round_probs:                      # HF casts the probabilities to the activation dtype before P.V
        att = _rnd(att, act_dtype)
    o = torch.matmul(att, v).transpose(1, 2).reshape(B, S, H)
    h = h + F.linear(_rnd(o, act_dtype), w[f"{p}.self_attn.o_proj.weight"].float())
    x = _rnd(rmsnorm(h, w[f"{p}.post_attention_layernorm.weight"], spec.rms_norm_eps), act_dtype)
    g = F.linear(x, w[f"{p}.mlp.gate_proj.weight"].float())
    u = F.linear(x, w[f"{p}.mlp.up_proj.weight"].float())
    h = h + F.linear(_rnd(F.silu(g) * u, act_dtype), w[f"{p}.mlp.down_proj.weight"].float())
    return h, (k, v)


def forward(w: Dict[str, torch.Tensor], spec: LlamaSpec, input_ids: torch.Tensor,
            audio_encodings: Union[None, torch.Tensor, Sequence[torch.Tensor]] = None,
            labels: Optional[torch.Tensor] = None, act_dtype=None, past_key_values=None, num_layers: Optional[int] = None,
            return_hidden: bool = False, round_probs: bool = True):
    """``WrappedLlamav2ForCausalLM.forward`` (no padding mask: full-length right-aligned batches).

    Returns dict(logits (B,S,V) fp32, loss or None, past_key_values, [hidden]).
    """
    B, S = input_ids.shape
    emb = w["model.embed_tokens.weight"].float()
    inputs_embeds = emb[input_ids]
    if audio_encodings is not None:
        pw, pb = w["model.mm_projector.weight"].float(), w["model.mm_projector.bias"].float()
        if isinstance(audio_encodings, (list, tuple)):
            feats = [F.linear(_rnd(a.float(), act_dtype), pw, pb) for a in audio_encodings]
        else:
            feats = F.linear(_rnd(audio_encodings.float(), act_dtype), pw, pb)
        inputs_embeds = splice_audio(input_ids, inputs_embeds, feats, spec, has_past=past_key_values is not None)
    past_len = 0 if past_key_values is None else past_key_values[0][0].shape[2]
    positions = torch.arange(past_len, past_len + S)
    cos, sin = rope_cos_sin(positions, spec.head_dim, spec.rope_theta)
    total = past_len + S
    mask = torch.full((S, total), float("-inf")).triu(diagonal=past_len + 1)[None, None]
    h = inputs_embeds.float()
    new_past = []
    L = spec.num_hidden_layers if num_layers is None else num_layers
    for i in range(L):
        h, kv = decoder_layer(h, w, i, spec, cos, sin, mask, act_dtype,
                              None if past_key_values is None else past_key_values[i], round_probs)
        new_past.append(kv)
    hn = _rnd(rmsnorm(h, w["model.norm.weight"], spec.rms_norm_eps), act_dtype)
    logits = F.linear(hn, w["lm_head.weight"].float())
    loss = None
    if labels is not None:
        shift_logits = logits[..., :-1, :].contiguous().view(-1, logits.shape[-1])
        shift_labels = labels[..., 1:].contiguous().view(-1)
        loss = F.cross_entropy(shift_logits, shift_labels, ignore_index=-100)
    out = dict(logits=logits, loss=loss, past_key_values=new_past)
    if return_hidden:
        out["hidden"] = h
    return out


def greedy_generate(w, spec: LlamaSpec, input_ids: torch.Tensor, audio_encodings, max_new_tokens: int,
                    act_dtype=None, eos_token_id: Optional[int] = None, round_probs: bool = True):
    """HF ``generate(do_sample=False)`` through ``prepare_inputs_for_generation``
    (m2t/models/llamav2.py:339-365): full prompt once, then one token per step with the KV cache;
    audio_encodings are only spliced on the first step (later steps contain no <audio_start>)."""
    out = forward(w, spec, input_ids, audio_encodings, act_dtype=act_dtype, round_probs=round_probs)
    past = out["past_key_values"]
    ids = input_ids
    for _ in range(max_new_tokens):
        nxt = out["logits"][:, -1].argmax(-1, keepdim=True)
        ids = torch.cat((ids, nxt), dim=1)
        if eos_token_id is not None and bool((nxt == eos_token_id).all()):
            break
        out = forward(w, spec, nxt, None, act_dtype=act_dtype, past_key_values=past, round_probs=round_probs)
        past = out["past_key_values"]
    return ids


def make_weights(spec: LlamaSpec, seed: int = 0, std: float = 0.02, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Random-init weights with HF state-dict names (``initializer_range`` 0.02; norms = 1)."""
    g = torch.Generator().manual_seed(seed)
    H, I, V = spec.hidden_size, spec.intermediate_size, spec.vocab_size

    def n(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    w = {"model.embed_tokens.weight": n(V, H), "model.norm.weight": (1 + 0.1 * torch.randn(H, generator=g)).to(dtype),
         "lm_head.weight": n(V, H), "model.mm_projector.weight": n(H, spec.mm_hidden_size),
         "model.mm_projector.bias": n(H)}
    for i in range(spec.num_hidden_layers):
        p = f"model.layers.{i}"
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            w[f"{p}.self_attn.{name}.weight"] = n(H, H)
        w[f"{p}.mlp.gate_proj.weight"] = n(I, H)
        w[f"{p}.mlp.up_proj.weight"] = n(I, H)
        w[f"{p}.mlp.down_proj.weight"] = n(H, I)
        w[f"{p}.input_layernorm.weight"] = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
        w[f"{p}.post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    return w
