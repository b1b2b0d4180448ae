"""MPT backbone on the HIP kernels: the arithmetic the reference reaches through
``WrappedMPTForCausalLM.forward`` (m2t/models/mpt.py:74-245,259-330 -> m2t/llava/model/mpt/modeling_mpt.py forward ->
blocks.py MPTBlock -> attention.py MultiheadAttention / scaled_multihead_dot_product_attention).

Per block:  LayerNorm -> fused Wqkv GEMM (-> clip_qkv clamp -> qk_ln LayerNorm over q and k) -> head split + KV-cache write
(the RoPE kernel with an identity rotation table) -> causal attention with the ALiBi bias (same flash / decode kernels as
Llama, slopes passed in) -> out_proj GEMM with the residual fused -> LayerNorm -> up_proj GEMM -> exact GELU -> down_proj
GEMM with the residual fused.  Final LayerNorm, logits against the TIED ``wte`` (x ``logit_scale``).  Head dim must be
128 (MPT-1B: 2048 / 16) -- the attention kernels are built for it.

Precision modes are those of the Llama engine: "split" (bf16 hi+lo operands, fp32-class, matches the reference's fp32 CPU
path) and "bf16" (single pass).  Weights are bf16 like ``model.to(bf16)`` of the reference recipe.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops


@dataclass
class MptDims:
    d_model: int = 2048
    n_heads: int = 16
    n_layers: int = 24
    expansion_ratio: int = 4
    vocab_size: int = 50432
    max_seq_len: int = 2048
    alibi_bias_max: int = 8
    qk_ln: bool = False
    clip_qkv: Optional[float] = None
    logit_scale: Optional[float] = None
    ln_eps: float = 1e-5
    mm_hidden_size: int = 512

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads


def alibi_slopes(n_heads: int, alibi_bias_max: int = 8) -> torch.Tensor:
    """m2t/llava/model/mpt/attention.py:462-469 (gen_slopes), as a flat fp32 [n_heads] vector."""
    n = 2 ** math.ceil(math.log2(n_heads))
    m = torch.arange(1, n + 1, dtype=torch.float32) * (alibi_bias_max / n)
    slopes = 1.0 / torch.pow(2, m)
    if n != n_heads:
        slopes = torch.cat([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes.contiguous()


class _Block:
    __slots__ = ("wqkv", "bqkv", "wo", "bo", "wup", "bup", "wdown", "bdown", "n1w", "n1b", "n2w", "n2b", "qlw", "qlb", "klw", "klb")


class HipMptEngine:
    def __init__(self, dims: MptDims, device="cuda", max_batch: int = 8, max_seq: int = 512, precision: str = "split"):
        if precision not in ("split", "bf16"):
            raise ValueError(f"precision must be 'split' or 'bf16', got {precision!r}")
        if dims.head_dim != 128:
            raise NotImplementedError(f"the HIP attention kernels are built for head_dim 128 (MPT-1B: 2048/16); got {dims.head_dim}")
        if dims.d_model % 64 or dims.mm_hidden_size % 32:
            raise NotImplementedError("d_model must be a multiple of 64 and mm_hidden_size of 32")
        self.dims, self.device = dims, torch.device(device)
        self.precision, self.split = precision, precision == "split"
        self.max_batch, self.smax = max_batch, ops.round_up(max_seq, 8)
        self.blocks: List[Optional[_Block]] = [None] * dims.n_layers
        self.wte = self.normf_w = self.normf_b = self.proj_w = self.proj_b = None
        self.slopes = alibi_slopes(dims.n_heads, dims.alibi_bias_max).to(self.device)
        self.fuse_decode_rope = os.environ.get("LLARK_DECODE_FUSE_ROPE", "1") != "0"      # head split + cache append inside the decode attention launch
        # identity rotation: the RoPE kernel then only splits heads, rounds to bf16 (hi/lo) and writes the KV cache
        self.cos = torch.ones((self.smax, 64), dtype=torch.float32, device=self.device)
        self.sin = torch.zeros((self.smax, 64), dtype=torch.float32, device=self.device)
        self._ws, self._ws_key = {}, None
        self.k_cache = self.vt_cache = self.k_cache_lo = self.vt_cache_lo = None
        self.cur_len = self.cur_batch = 0

    # ---- weights -------------------------------------------------------------------------------
    def _bf16(self, t):
        return None if t is None else t.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()

    def _f32(self, t):
        return None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Reference names: ``transformer.wte.weight``, ``transformer.blocks.N.{norm_1,norm_2}.{weight,bias}``,
        ``...attn.{Wqkv,out_proj,q_ln,k_ln}.*``, ``...ffn.{up_proj,down_proj}.*``, ``transformer.norm_f.*``,
        ``transformer.mm_projector.*`` (biases optional: ``no_bias`` models have none)."""
        d = self.dims
        for i in range(d.n_layers):
            p = f"transformer.blocks.{i}"
            B = _Block()
            B.wqkv, B.bqkv = self._bf16(sd[f"{p}.attn.Wqkv.weight"]), self._f32(sd.get(f"{p}.attn.Wqkv.bias"))
            B.wo, B.bo = self._bf16(sd[f"{p}.attn.out_proj.weight"]), self._f32(sd.get(f"{p}.attn.out_proj.bias"))
            B.wup, B.bup = self._bf16(sd[f"{p}.ffn.up_proj.weight"]), self._f32(sd.get(f"{p}.ffn.up_proj.bias"))
            B.wdown, B.bdown = self._bf16(sd[f"{p}.ffn.down_proj.weight"]), self._f32(sd.get(f"{p}.ffn.down_proj.bias"))
            B.n1w, B.n1b = self._f32(sd[f"{p}.norm_1.weight"]), self._f32(sd.get(f"{p}.norm_1.bias"))
            B.n2w, B.n2b = self._f32(sd[f"{p}.norm_2.weight"]), self._f32(sd.get(f"{p}.norm_2.bias"))
            if d.qk_ln:
                B.qlw, B.qlb = self._f32(sd[f"{p}.attn.q_ln.weight"]), self._f32(sd.get(f"{p}.attn.q_ln.bias"))
                B.klw, B.klb = self._f32(sd[f"{p}.attn.k_ln.weight"]), self._f32(sd.get(f"{p}.attn.k_ln.bias"))
            else:
                B.qlw = B.qlb = B.klw = B.klb = None
            for t in (B.wqkv, B.wo, B.wup, B.wdown):
                ops.attach_frag(t, t.shape[0])
            self.blocks[i] = B
        self.wte = self._bf16(sd["transformer.wte.weight"])
        ops.attach_frag(self.wte, self.wte.shape[0])
        self.normf_w, self.normf_b = self._f32(sd["transformer.norm_f.weight"]), self._f32(sd.get("transformer.norm_f.bias"))
        if "transformer.mm_projector.weight" in sd:
            self.proj_w, self.proj_b = self._bf16(sd["transformer.mm_projector.weight"]), self._f32(sd["transformer.mm_projector.bias"])

    # ---- buffers -------------------------------------------------------------------------------
    def _workspace(self, batch: int, s: int):
        if self._ws_key != (batch, s):
            d, dev = self.dims, self.device
            rows, D, E = batch * s, d.d_model, d.expansion_ratio * d.d_model
            f32, bf = dict(dtype=torch.float32, device=dev), dict(dtype=torch.bfloat16, device=dev)
            ws = {"h": torch.empty((rows, D), **f32), "qkv": torch.empty((rows, 3 * D), **f32), "up": torch.empty((rows, E), **f32),
                  "x16": torch.empty((rows, D), **bf), "q": torch.empty((batch, d.n_heads, s, 128), **bf),
                  "att": torch.empty((rows, D), **bf), "act": torch.empty((rows, E), **bf)}
            for name in ("x16", "q", "att", "act"):
                ws[name + "_lo"] = torch.empty_like(ws[name]) if self.split else None
            self._ws, self._ws_key = ws, (batch, s)
        return self._ws

    def reset(self, batch: int) -> None:
        d = self.dims
        if self.k_cache is None or self.k_cache.shape[1] != batch:
            ks = (d.n_layers, batch, d.n_heads, self.smax, 128)
            vs = (d.n_layers, batch, d.n_heads, 128, self.smax)
            z = lambda shp: torch.zeros(shp, dtype=torch.bfloat16, device=self.device)
            self.k_cache, self.vt_cache = z(ks), z(vs)
            self.k_cache_lo, self.vt_cache_lo = (z(ks), z(vs)) if self.split else (None, None)
        self.cur_len, self.cur_batch = 0, batch

    # ---- forward -------------------------------------------------------------------------------
    def forward_tokens(self, input_ids: torch.Tensor, audio_segments: Sequence[Tuple[int, int, torch.Tensor]] = (), pos0: int = 0,
                       last_only: bool = False, return_hidden: bool = False, num_layers: Optional[int] = None) -> torch.Tensor:
        """input_ids (B,S) int64 on device; audio_segments as in HipLlamaEngine.forward_tokens (projected frames overwrite
        rows start+1 .. start+F).  Returns fp32 logits (B,S,V) (or (B,1,V) if last_only).  KV positions pos0..pos0+S-1 written."""
        d = self.dims
        assert self.wte is not None and all(b is not None for b in self.blocks), "weights not loaded"
        B, S = input_ids.shape
        if pos0 == 0:
            self.reset(B)
        assert B == self.cur_batch and pos0 == self.cur_len, "KV cache is out of sync with the requested positions"
        if pos0 + S > min(self.smax, d.max_seq_len):
            raise ValueError(f"Cannot forward input with past sequence length {pos0} and current sequence length {S}: "
                             f"this engine holds {min(self.smax, d.max_seq_len)} positions")
        ws = self._workspace(B, S)
        h, D, E, nh = ws["h"], d.d_model, d.expansion_ratio * d.d_model, d.n_heads
        sp = self.split
        ops.embed_gather(input_ids.reshape(-1).contiguous(), self.wte, h)
        for (b, start, frames) in audio_segments:
            assert self.proj_w is not None, "mm_projector weights not loaded"
            F = frames.shape[0]
            a16, a16_lo = ops.split16(frames.contiguous(), torch.bfloat16, want_lo=sp)
            r0 = b * S + start + 1
            ops.gemm16(a16, a16_lo, self.proj_w, self.proj_b, D, ops.EPI_F32, c=h[r0: r0 + F])
        n_layers = d.n_layers if num_layers is None else num_layers
        for i in range(n_layers):
            Bk = self.blocks[i]
            kc, vc = self.k_cache[i], self.vt_cache[i]
            kcl, vcl = (self.k_cache_lo[i], self.vt_cache_lo[i]) if sp else (None, None)
            ops.layernorm_bf16(h, Bk.n1w, Bk.n1b, d.ln_eps, ws["x16"], ws["x16_lo"])
            ops.gemm16(ws["x16"], ws["x16_lo"], Bk.wqkv, Bk.bqkv, 3 * D, ops.EPI_F32, c=ws["qkv"])
            if d.clip_qkv:
                ops.clamp_f32_(ws["qkv"], d.clip_qkv)
            if d.qk_ln:
                ops.layernorm_f32_(ws["qkv"][:, :D], Bk.qlw, Bk.qlb, d.ln_eps)
                ops.layernorm_f32_(ws["qkv"][:, D: 2 * D], Bk.klw, Bk.klb, d.ln_eps)
            if S == 1 and self.fuse_decode_rope:
                ops.attn_decode_rope(ws["qkv"], B, nh, 128, pos0, self.cos, self.sin, kc, vc, ws["att"], kcl, vcl, ws["att_lo"],
                                     alibi_slopes=self.slopes)
            else:
                ops.rope_split_heads(ws["qkv"], B, S, nh, 128, pos0, self.cos, self.sin, ws["q"], kc, vc, ws["q_lo"], kcl, vcl)
            if S == 1 and self.fuse_decode_rope:
                pass
            elif S == 1:
                ops.attn_decode(ws["q"], kc, vc, B, nh, 128, pos0 + 1, ws["att"], ws["q_lo"], kcl, vcl, ws["att_lo"], alibi_slopes=self.slopes)
            else:
                ops.attn_prefill(ws["q"], kc, vc, B, S, nh, 128, pos0, ws["att"], ws["q_lo"], kcl, vcl, ws["att_lo"], alibi_slopes=self.slopes)
            ops.gemm16(ws["att"], ws["att_lo"], Bk.wo, Bk.bo, D, ops.EPI_RESID, c=h, resid=h)
            ops.layernorm_bf16(h, Bk.n2w, Bk.n2b, d.ln_eps, ws["x16"], ws["x16_lo"])
            ops.gemm16(ws["x16"], ws["x16_lo"], Bk.wup, Bk.bup, E, ops.EPI_F32, c=ws["up"])
            ops.gelu_split_bf16(ws["up"], ws["act"], ws["act_lo"])
            ops.gemm16(ws["act"], ws["act_lo"], Bk.wdown, Bk.bdown, D, ops.EPI_RESID, c=h, resid=h)
        self.cur_len = pos0 + S
        if return_hidden:
            return h.view(B, S, D)
        if last_only and S > 1:
            hl = h.view(B, S, D)[:, -1].contiguous()
            x16 = torch.empty((B, D), dtype=torch.bfloat16, device=self.device)
            x16_lo = torch.empty_like(x16) if sp else None
            ops.layernorm_bf16(hl, self.normf_w, self.normf_b, d.ln_eps, x16, x16_lo)
            rows = B
        else:
            x16, x16_lo = ws["x16"], ws["x16_lo"]
            ops.layernorm_bf16(h, self.normf_w, self.normf_b, d.ln_eps, x16, x16_lo)
            rows = B * S
        logits = torch.empty((rows, d.vocab_size), dtype=torch.float32, device=self.device)
        ops.gemm16(x16, x16_lo, self.wte, None, d.vocab_size, ops.EPI_F32, c=logits)
        if d.logit_scale is not None:
            ops.scale_f32_(logits, float(d.logit_scale))
        return logits.view(B, -1, d.vocab_size)
