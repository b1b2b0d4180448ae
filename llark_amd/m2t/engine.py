"""Native Llama-2 execution engine on the HIP kernels (weights in kernel layout, KV cache, workspaces).

This is what runs underneath :class:`llark_amd.m2t.llamav2.WrappedLlamav2ForCausalLM`; it replaces
the arithmetic the reference delegates to HF ``LlamaModel.forward`` + ``lm_head`` + ``mm_projector``
(m2t/models/llamav2.py:124,133,224-234,312).

HBM layout:
  embed      bf16 [V][H]                      lm_head  bf16 [V][H]
  per layer  wqkv bf16 [3H][H] (q|k|v rows)   wo bf16 [H][H]
             wgu  bf16 [2I][H]  rows interleaved in blocks of 32: [gate 32 | up 32] (SwiGLU epilogue)
             wdown bf16 [H][I]                norms fp32 [H]
  projector  bf16 [H][mm] + fp32 bias
  h          fp32 [B*S][H] residual stream (updated in place by GEMM epilogues)
  k_cache    bf16 [L][B][nh][Smax][128]       vt_cache bf16 [L][B][nh][128][Smax]  (V transposed)
"""
from __future__ import annotations

from dataclasses import dataclass
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops


@dataclass
class LlamaDims:
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    vocab_size: int = 32000
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    mm_hidden_size: int = 4800
    max_position_embeddings: int = 4096

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[I][H] gate, up -> [2I][H] with rows [gate 0..31 | up 0..31 | gate 32..63 | up 32..63 | ...]."""
    I, H = gate.shape
    assert I % 32 == 0, "intermediate_size must be a multiple of 32"
    return torch.stack((gate.view(I // 32, 32, H), up.view(I // 32, 32, H)), dim=1).reshape(2 * I, H).contiguous()


class _Layer:
    __slots__ = ("wqkv", "wo", "wgu", "wdown", "ln1", "ln2", "wqkv_rope")


class HipLlamaEngine:
    """precision: "split" (default) = fp32-class: every Linear input, q/k/v and the attention output are
    carried as bf16 hi+lo planes (16 significant bits) and every GEMM / attention product runs the extra
    MFMA passes -- logits then match the reference's fp32 CPU path (bf16-valued weights) to ~1e-5;
    "bf16" = the reference's GPU dtype flow (single bf16 rounding at those points), ~2x fewer MFMAs."""

    def __init__(self, dims: LlamaDims, device="cuda", max_batch: int = 8, max_seq: int = 512, precision: str = "split", frag_weights: bool = True):
        """frag_weights: keep a fragment-major twin of every Linear weight for the B-direct GEMM (inference; the
        training step updates the row-major weights in place and therefore builds its engine with False)."""
        self.frag_weights = frag_weights
        if precision not in ("split", "bf16"):
            raise ValueError(f"precision must be 'split' or 'bf16', got {precision!r}")
        self.precision = precision
        self.split = precision == "split"
        if dims.head_dim != 128:
            raise NotImplementedError(f"the HIP attention kernels are built for head_dim 128 (Llama-2); got {dims.head_dim}")
        if dims.intermediate_size % 32 or dims.hidden_size % 32 or dims.mm_hidden_size % 32:
            raise NotImplementedError("hidden / intermediate / mm_hidden sizes must be multiples of 32")
        self.dims = dims
        self.device = torch.device(device)
        self.max_batch = max_batch
        self.smax = ops.round_up(max_seq, 8)
        self.layers: List[Optional[_Layer]] = [None] * dims.num_hidden_layers
        self.embed = self.lm_head = self.norm = self.proj_w = self.proj_b = None
        # RoPE tables exactly as HF computes them (fp32, inv_freq = theta^(-2i/d), emb = cat(freqs, freqs))
        hd = dims.head_dim
        inv_freq = 1.0 / (dims.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        freqs = torch.arange(self.smax, dtype=torch.float32)[:, None] * inv_freq[None, :]
        self.cos = freqs.cos().contiguous().to(self.device)
        self.sin = freqs.sin().contiguous().to(self.device)
        self._ws: Dict[str, torch.Tensor] = {}
        self._ws_key = None
        self.k_cache = self.vt_cache = self.k_cache_lo = self.vt_cache_lo = None
        self.cur_len = 0
        self.cur_batch = 0
        # graph-captured decode step (one hipGraph per batch size; position read from device memory)
        # o_proj / down_proj + following RMSNorm in one launch ("last workgroup done" tail).  Bit-identical, but OFF by
        # default.  Round 1 fenced with __threadfence() in every workgroup (an L2 write-back + invalidate each on the 8-XCD
        # part: ~65 us per launch, 10.1 vs 5.9 ms per decode step); with write-through stores + a relaxed ticket + ONE
        # acquire in the last workgroup (round 2) the tail is cheap but still serial: 4.34 vs 3.97 ms per token at B = 1 --
        # the last workgroup normalises alone while the chip idles, which costs more than the 5 us rmsnorm launch it removes.
        self.fuse_decode_norm = os.environ.get("LLARK_DECODE_FUSE_NORM", "0") == "1"
        # RMSNorm fused INTO the consuming decode GEMM (each workgroup re-derives the row scales): bit-identical, saves
        # two launches per layer (+ the final norm before lm_head).  Measured: B = 8 split 5.89 vs 6.04 ms per step, but
        # B = 8 single-pass 5.26 vs 4.85 ms and B = 1 split 492 vs 480 ms per 64-token generate.  Why it cannot win: every
        # one of the ~768 workgroups of a decode GEMM re-normalises and re-splits its activation fragments (~50 VALU
        # instructions per k-step and wave), so the fused kernel turns VALU-bound (+15 us per launch, more than the
        # rmsnorm launch it removes); hoisting the weight loads above the scale derivation and batching the raw loads
        # before the conversions (tried, round 1) changed nothing.  Opt-in (LLARK_DECODE_FUSE_NORM_A=1).
        # Round 3: the LDS-DMA streaming Linear (csrc/gemv_dma.hip) holds the whole activation row in every wave's registers, so
        # it derives rstd and the hi / lo planes itself under the weight stream that is already flowing: that form (B = 1, weights
        # >= 64 MB: q/k/v, gate/up, lm_head) is the default ("auto"); "1" forces the fusion for every decode shape (the MFMA skinny
        # kernel's per-k-step form for the others), "0" switches it off.
        self.fuse_decode_norm_a = os.environ.get("LLARK_DECODE_FUSE_NORM_A", "auto")      # property: also takes True / False
        self.decode_graph = os.environ.get("LLARK_DECODE_GRAPH", "0") == "1"      # measured: no gain on ROCm 7.2 (kernel boundaries remain), opt-in
        # decode step as a recorded host launch list over static buffers (ops.LaunchList): removes the per-launch Python cost
        self.decode_replay = os.environ.get("LLARK_DECODE_REPLAY", "0") == "1"
        # decode: RoPE + KV-cache append inside the attention launch (one launch per layer fewer); LLARK_DECODE_FUSE_ROPE=0 = two launches
        self.fuse_decode_rope = os.environ.get("LLARK_DECODE_FUSE_ROPE", "1") != "0"
        # round 6: batch-1 decode with o_proj launched on a side stream while the attention still runs (it fills its weight ring and spins on
        # the attention's arrival counter) and gate/up filling its ring while o_proj runs (llark_gemv16_dma_chain); opt-in until measured
        self.decode_chain = os.environ.get("LLARK_DECODE_CHAIN", "0") == "1"
        self._chain = None
        # prefill: RoPE + head split + K / V^T cache writes inside the q|k|v GEMM's epilogue (llark_gemm16_fragw_rope_qkv: no fp32 qkv
        # tensor, no rope_split_kernel launch).  "auto" (default) = fused wherever the two-launch path runs the same whole-tile kernel,
        # so q / K / V^T and the logits stay BIT-equal (tests/test_llama_gpu.py); "1" = fused for every prefill of >= 32 positions;
        # "0" = two launches.  Measured at 7B, 8 x 371 (profiles/r04_rope_fuse_ab_v2.txt, same engine, interleaved): forward 42.48 ->
        # 40.97 ms bf16, 79.03 -> 77.82 ms split.  (The first version stored the V tiles straight from the accumulator lanes -- 64
        # two-byte pieces on 64 cache lines per instruction -- and lost what the removed launch gained: 42.63 -> 42.24 / 79.10 -> 80.31,
        # profiles/r04_rope_fuse_ab_v1.txt; the V tiles now go through a wave-private LDS transposition.)  Costs a second fragment-major
        # copy of the q|k|v weight in the epilogue's row order (+ 3 H^2 x 2 bytes per layer: 3.2 GB at 7B), built at load time unless "0".
        self.fuse_prefill_rope = os.environ.get("LLARK_PREFILL_FUSE_ROPE", "auto")
        if self.fuse_prefill_rope not in ("0", "1", "auto"):
            raise ValueError(f"LLARK_PREFILL_FUSE_ROPE must be 0, 1 or auto, got {self.fuse_prefill_rope!r}")
        # prefill of an even batch as TWO half-batches on two HIP streams (round 5; LLARK_PREFILL_STREAMS=2, default 1 until measured):
        # every kernel of the layer stack leaves part of the chip idle in its last round of tiles (M = 8 x 371 = 2968 rows: q|k|v 2.25
        # rounds of the 512 resident workgroups, o_proj / down_proj 0.75) and none of them overlaps its successor on ONE stream; two
        # independent half-batches let the hardware dispatcher fill the tail of one stream's kernel with the other stream's workgroups.
        # Same kernels, same per-row arithmetic; the K cuts of o_proj / down_proj follow the half-batch's tile count (see ops.gemm16_fragw).
        self.prefill_streams = int(os.environ.get("LLARK_PREFILL_STREAMS", "1"))
        if self.prefill_streams not in (1, 2):
            raise ValueError(f"LLARK_PREFILL_STREAMS must be 1 or 2, got {self.prefill_streams}")
        self._side_streams = None
        self._resident_wgs = None
        self._rope_rows = None                                      # gather index of the epilogue's q|k|v row order (built on first use)
        self._dec: Dict[int, dict] = {}

    # ---- weights -------------------------------------------------------------------------------
    def _bf16(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def set_layer(self, i: int, q, k, v, o, gate, up, down, ln1, ln2) -> None:
        L = _Layer()
        L.wqkv = torch.cat((self._bf16(q), self._bf16(k), self._bf16(v)), dim=0).contiguous()
        L.wo = self._bf16(o)
        L.wgu = interleave_gate_up(self._bf16(gate), self._bf16(up))
        L.wdown = self._bf16(down)
        L.ln1, L.ln2 = self._f32(ln1), self._f32(ln2)
        self._dec.clear()                                     # captured graphs hold the old weight pointers
        old = self.layers[i]
        if old is not None:
            for t in (old.wqkv, old.wo, old.wgu, old.wdown):
                ops.detach_frag(t)
        if self.frag_weights:
            for t in (L.wqkv, L.wo, L.wgu, L.wdown):
                ops.attach_frag(t, t.shape[0])
        L.wqkv_rope = None                                    # built on the first prefill that takes the fused epilogue (_rope_weight)
        self.layers[i] = L

    def _rope_capable(self, L) -> bool:
        # only next to the fragment-major twin the two-launch path uses (LLARK_FRAG=0 switches both off: same kernel family either way)
        return (self.frag_weights and hasattr(L.wqkv, "_llark_frag") and self.fuse_prefill_rope != "0" and self.dims.num_attention_heads % 2 == 0
                and self.dims.hidden_size % 64 == 0)

    def _rope_weight(self, L):
        """The q|k|v weight in the fused epilogue's row order, fragment-major: + 3 H^2 x 2 bytes per layer (3.2 GB at 7B), so it is
        built LAZILY -- engines that only decode, prefill short prompts or train never pay for it."""
        if L.wqkv_rope is None and self._rope_capable(L):
            try:
                if self._rope_rows is None:
                    self._rope_rows = ops.rope_qkv_row_order(self.dims.num_attention_heads, self.dims.head_dim).to(self.device)
                L.wqkv_rope = ops.pack_weight16_frag(L.wqkv.index_select(0, self._rope_rows).contiguous(), L.wqkv.shape[0])
            except torch.cuda.OutOfMemoryError:
                # built inside the first qualifying prefill, i.e. after the KV cache and the activations were allocated: running out of
                # memory here must not fail the forward -- this engine keeps the two-launch path from now on
                self.fuse_prefill_rope = "0"
                for M in self.layers:
                    if M is not None:
                        M.wqkv_rope = None
                torch.cuda.empty_cache()
                return None
        return L.wqkv_rope

    def prepare_prefill(self) -> bool:
        """Build the fused-RoPE q|k|v twins of every layer NOW (+ 3 H^2 x 2 bytes per layer, 3.2 GB at 7B) instead of inside the first
        qualifying prefill: callers that time their first forward, or want an out-of-memory condition at load time rather than in the
        middle of inference, call this after loading the weights.  Returns False when this engine does not take the fused epilogue."""
        return all(L is not None and self._rope_weight(L) is not None for L in self.layers)

    def set_globals(self, embed, norm, lm_head, proj_w=None, proj_b=None) -> None:
        self._dec.clear()
        self.embed, self.norm, self.lm_head = self._bf16(embed), self._f32(norm), self._bf16(lm_head)
        if proj_w is not None:
            self.proj_w, self.proj_b = self._bf16(proj_w), self._f32(proj_b)
        if self.frag_weights:
            ops.attach_frag(self.lm_head, self.lm_head.shape[0])

    def state_dict_hf(self) -> Dict[str, torch.Tensor]:
        """Weights under the reference's state-dict names and layouts (inverse of :meth:`load_state_dict`): the fused
        q|k|v rows are split, gate/up rows de-interleaved.  Tensors are views / copies on the engine's device."""
        d = self.dims
        H, I = d.hidden_size, d.intermediate_size
        sd: Dict[str, torch.Tensor] = {}
        for i, L in enumerate(self.layers):
            p = f"model.layers.{i}"
            sd[f"{p}.self_attn.q_proj.weight"], sd[f"{p}.self_attn.k_proj.weight"], sd[f"{p}.self_attn.v_proj.weight"] = (
                L.wqkv[:H], L.wqkv[H: 2 * H], L.wqkv[2 * H:])
            sd[f"{p}.self_attn.o_proj.weight"] = L.wo
            gu = L.wgu.view(I // 32, 2, 32, H)
            sd[f"{p}.mlp.gate_proj.weight"] = gu[:, 0].reshape(I, H)
            sd[f"{p}.mlp.up_proj.weight"] = gu[:, 1].reshape(I, H)
            sd[f"{p}.mlp.down_proj.weight"] = L.wdown
            sd[f"{p}.input_layernorm.weight"] = L.ln1
            sd[f"{p}.post_attention_layernorm.weight"] = L.ln2
        sd["model.embed_tokens.weight"], sd["model.norm.weight"], sd["lm_head.weight"] = self.embed, self.norm, self.lm_head
        if self.proj_w is not None:
            sd["model.mm_projector.weight"], sd["model.mm_projector.bias"] = self.proj_w, self.proj_b
        return sd

    def drop_frag_weights(self) -> None:
        """Called by the training step: the row-major weights are about to be updated in place."""
        self.frag_weights = False
        for L in self.layers:
            if L is not None:
                for t in (L.wqkv, L.wo, L.wgu, L.wdown):
                    ops.detach_frag(t)
                L.wqkv_rope = None
        if self.lm_head is not None:
            ops.detach_frag(self.lm_head)

    def load_state_dict(self, sd) -> None:
        """HF / reference state-dict names (model.layers.N.self_attn.q_proj.weight, ..., model.mm_projector.*)."""
        d = self.dims
        for i in range(d.num_hidden_layers):
            p = f"model.layers.{i}"
            self.set_layer(i, sd[f"{p}.self_attn.q_proj.weight"], sd[f"{p}.self_attn.k_proj.weight"],
                           sd[f"{p}.self_attn.v_proj.weight"], sd[f"{p}.self_attn.o_proj.weight"],
                           sd[f"{p}.mlp.gate_proj.weight"], sd[f"{p}.mlp.up_proj.weight"], sd[f"{p}.mlp.down_proj.weight"],
                           sd[f"{p}.input_layernorm.weight"], sd[f"{p}.post_attention_layernorm.weight"])
        self.set_globals(sd["model.embed_tokens.weight"], sd["model.norm.weight"], sd["lm_head.weight"],
                         sd.get("model.mm_projector.weight"), sd.get("model.mm_projector.bias"))

    # ---- workspaces ----------------------------------------------------------------------------
    def _workspace(self, batch: int, s: int):
        key = (batch, s)
        if self._ws_key != key:
            d, dev = self.dims, self.device
            rows = batch * s
            H, I = d.hidden_size, d.intermediate_size
            ws = {
                "h": torch.empty((rows, H), dtype=torch.float32, device=dev),
                "x16": torch.empty((rows, H), dtype=torch.bfloat16, device=dev),
                "qkv": torch.empty((rows, 3 * H), dtype=torch.float32, device=dev),
                "q": torch.empty((batch, d.num_attention_heads, s, d.head_dim), dtype=torch.bfloat16, device=dev),
                "att": torch.empty((rows, H), dtype=torch.bfloat16, device=dev),
                "act": torch.empty((rows, I), dtype=torch.bfloat16, device=dev),
            }
            if self.split:
                for name in ("x16", "q", "att", "act"):
                    ws[name + "_lo"] = torch.empty_like(ws[name])
            else:
                for name in ("x16", "q", "att", "act"):
                    ws[name + "_lo"] = None
            self._ws, self._ws_key = ws, key
        return self._ws

    def set_precision(self, precision: str) -> None:
        """Switch between the "split" (fp32-class) and "bf16" activation flows.  The weights are the same bf16 values in both;
        activation planes, the KV cache and captured decode graphs are rebuilt on the next call."""
        if precision not in ("split", "bf16"):
            raise ValueError(f"precision must be 'split' or 'bf16', got {precision!r}")
        if precision == self.precision:
            return
        self.precision, self.split = precision, precision == "split"
        self._ws, self._ws_key = None, None
        self._dec.clear()
        self.k_cache = self.vt_cache = self.k_cache_lo = self.vt_cache_lo = None

    def _ensure_cache(self, batch: int):
        d = self.dims
        if self.k_cache is None or self.k_cache.shape[1] < batch:
            self._dec.clear()                                 # captured graphs hold the old cache pointers
            shape_k = (d.num_hidden_layers, batch, d.num_attention_heads, self.smax, d.head_dim)
            shape_v = (d.num_hidden_layers, batch, d.num_attention_heads, d.head_dim, self.smax)
            self.k_cache = torch.zeros(shape_k, dtype=torch.bfloat16, device=self.device)
            self.vt_cache = torch.zeros(shape_v, dtype=torch.bfloat16, device=self.device)
            self.k_cache_lo = torch.zeros(shape_k, dtype=torch.bfloat16, device=self.device) if self.split else None
            self.vt_cache_lo = torch.zeros(shape_v, dtype=torch.bfloat16, device=self.device) if self.split else None

    def _prefill_rope_fused(self, batch: int, s: int) -> bool:
        """Does this prefill take the q|k|v product with RoPE in its epilogue?  "auto" = only where llark_gemm16_fragw would run the
        same whole 128x256 tiles (the library answers: llark_gemm16_fragw_whole_tiles), so results stay bit-equal."""
        if self.fuse_prefill_rope == "0" or s < 32 or batch * s < ops.FRAG_MIN_ROWS:
            return False
        if self.fuse_prefill_rope == "1":
            return True
        H = self.dims.hidden_size
        return ops.gemm16_fragw_whole_tiles(self.split, ops.EPI_F32, batch * s, 3 * H, H)

    # ---- forward -------------------------------------------------------------------------------
    def _layers_forward(self, ws, batch: int, s: int, pos0: int, num_layers: Optional[int] = None, pos_dev=None, hidden_sink=None) -> bool:
        """Runs the decoder layers on ws["h"].  Returns True when ws["x16"] already holds RMSNorm_final(h) (the fused
        decode path normalises inside the producing GEMM), False when the caller still has to apply the final norm."""
        d = self.dims
        H, I, nh, hd = d.hidden_size, d.intermediate_size, d.num_attention_heads, d.head_dim
        h = ws["h"]
        n_layers = d.num_hidden_layers if num_layers is None else num_layers
        sp = self.split
        # decode (one token per sequence): o_proj / down_proj carry the FOLLOWING RMSNorm in their launch
        fused = s == 1 and batch <= 16 and n_layers > 0 and self.fuse_decode_norm and H <= 8192 and hidden_sink is None
        norm_a = (s == 1 and batch <= 16 and not fused and H % 32 == 0 and
                  (self.fuse_decode_norm_a == "1" or (self.fuse_decode_norm_a == "auto" and ops.gemv_dma_rmsnorm_takes(batch, 3 * H, H))))
        if fused:
            ops.rmsnorm_bf16(h, self.layers[0].ln1, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
        rope_fused = s > 1 and pos_dev is None and self._prefill_rope_fused(batch, s)
        if rope_fused:
            rope_fused = all(self._rope_weight(self.layers[i]) is not None for i in range(n_layers))
        if (self.prefill_streams == 2 and rope_fused and batch >= 2 and batch % 2 == 0 and self._prefill_rope_fused(batch // 2, s)
                and hidden_sink is None and not torch.cuda.is_current_stream_capturing()):
            self._prefill_two_streams(ws, batch, s, pos0, n_layers)
            return False
        if (self.decode_chain and s == 1 and batch == 1 and norm_a and self.fuse_decode_rope and pos_dev is None and hidden_sink is None
                and not fused and H <= 4096 and not torch.cuda.is_current_stream_capturing()):
            self._decode_layers_chained(ws, pos0, n_layers)
            return False
        for i in range(n_layers):
            L = self.layers[i]
            if hidden_sink is not None:                       # HF output_hidden_states: the stream as it ENTERS every layer
                hidden_sink.append(h.view(batch, s, H).clone())
            kc, vc = self.k_cache[i, :batch], self.vt_cache[i, :batch]
            kcl = self.k_cache_lo[i, :batch] if sp else None
            vcl = self.vt_cache_lo[i, :batch] if sp else None
            if not kc.is_contiguous():            # batch smaller than the allocated cache
                raise ops._lib.LlarkHipError("KV cache batch mismatch: call reset(batch) before prefill")
            if rope_fused:                           # prefill: RoPE / head split / cache writes in the q|k|v epilogue
                ops.rmsnorm_bf16(h, L.ln1, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
                ops.gemm16_fragw_rope_qkv(ws["x16"], ws["x16_lo"], L.wqkv_rope, H, batch, s, nh, pos0, self.cos, self.sin, ws["q"], kc, vc,
                                          ws["q_lo"], kcl, vcl)
            elif norm_a:                             # decode: RMSNorm fused into the consuming weight-streaming GEMM
                ops.gemm16_rmsnorm_a(h, L.ln1, d.rms_norm_eps, L.wqkv, 3 * H, ops.EPI_F32, sp, c=ws["qkv"])
            else:
                if not fused:
                    ops.rmsnorm_bf16(h, L.ln1, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
                ops.gemm16(ws["x16"], ws["x16_lo"], L.wqkv, None, 3 * H, ops.EPI_F32, c=ws["qkv"])
            if rope_fused:
                ops.attn_prefill(ws["q"], kc, vc, batch, s, nh, hd, pos0, ws["att"], ws["q_lo"], kcl, vcl, ws["att_lo"])
            elif s == 1 and self.fuse_decode_rope:
                ops.attn_decode_rope(ws["qkv"], batch, nh, hd, pos_dev if pos_dev is not None else pos0, self.cos, self.sin, kc, vc,
                                     ws["att"], kcl, vcl, ws["att_lo"])
            elif pos_dev is not None:               # decode step, position in device memory (graph-capturable)
                ops.rope_split_heads_dpos(ws["qkv"], batch, nh, hd, pos_dev, self.cos, self.sin, ws["q"], kc, vc,
                                          ws["q_lo"], kcl, vcl)
                ops.attn_decode_dpos(ws["q"], kc, vc, batch, nh, hd, pos_dev, ws["att"], ws["q_lo"], kcl, vcl, ws["att_lo"])
            else:
                ops.rope_split_heads(ws["qkv"], batch, s, nh, hd, pos0, self.cos, self.sin, ws["q"], kc, vc,
                                     ws["q_lo"], kcl, vcl)
                if s == 1:
                    ops.attn_decode(ws["q"], kc, vc, batch, nh, hd, pos0 + 1, ws["att"], ws["q_lo"], kcl, vcl, ws["att_lo"])
                else:
                    ops.attn_prefill(ws["q"], kc, vc, batch, s, nh, hd, pos0, ws["att"], ws["q_lo"], kcl, vcl, ws["att_lo"])
            if fused:
                ops.gemm16_resid_rmsnorm(ws["att"], ws["att_lo"], L.wo, h, L.ln2, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
            else:
                ops.gemm16(ws["att"], ws["att_lo"], L.wo, None, H, ops.EPI_RESID, c=h, resid=h)
                if not norm_a:
                    ops.rmsnorm_bf16(h, L.ln2, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
            if norm_a:
                ops.gemm16_rmsnorm_a(h, L.ln2, d.rms_norm_eps, L.wgu, 2 * I, ops.EPI_SWIGLU_SPLIT if sp else ops.EPI_SWIGLU16, sp,
                                     out_hi=ws["act"], out_lo=ws["act_lo"])
            else:
                ops.gemm16(ws["x16"], ws["x16_lo"], L.wgu, None, 2 * I, ops.EPI_SWIGLU_SPLIT if sp else ops.EPI_SWIGLU16,
                           out_hi=ws["act"], out_lo=ws["act_lo"])
            if fused:
                nxt = self.layers[i + 1].ln1 if i + 1 < n_layers else self.norm
                ops.gemm16_resid_rmsnorm(ws["act"], ws["act_lo"], L.wdown, h, nxt, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
            else:
                ops.gemm16(ws["act"], ws["act_lo"], L.wdown, None, H, ops.EPI_RESID, c=h, resid=h)
        return fused

    def _decode_layers_chained(self, ws, pos0: int, n_layers: int) -> None:
        """The decoder layers of a batch-1 decode step with two launches per layer overlapping their producers (round 6):

            main stream:  q|k|v (RMSNorm fused) -> attention [signals A_i] ------------> gate/up [fills its ring, waits O_i] -> down_proj
            side stream:            (after q|k|v)  o_proj [fills its ring, waits A_i, signals O_i]

        o_proj's 33.5 MB of weights start streaming into LDS while the attention launch -- a chain of memory latencies on 32 of 256 CUs, no
        weight traffic -- is running, and gate/up's while o_proj streams.  Both chained kernels use a 64 KiB ring so that two workgroups share a
        CU (98 / 144 registers: the only pairs of the layer that fit 4 waves per SIMD: DESIGN.md 7(d)).  Counters are monotonic device words.
        Same kernels, same arithmetic: results equal the unchained path."""
        d = self.dims
        H, I, nh, hd = d.hidden_size, d.intermediate_size, d.num_attention_heads, d.head_dim
        sp = self.split
        if self._chain is None or self._chain["layers"] != n_layers:
            self._chain = {"layers": n_layers, "cnt": torch.zeros((2 * n_layers,), dtype=torch.int32, device=self.device), "epoch": 0,
                           "side": torch.cuda.Stream(device=self.device), "ev": torch.cuda.Event(), "blocks_o": ops.gemv16_dma_blocks(ops.EPI_RESID, H)}
        ch = self._chain
        ch["epoch"] += 1
        ep, cnt, side, ev = ch["epoch"], ch["cnt"], ch["side"], ch["ev"]
        main = torch.cuda.current_stream()
        h = ws["h"]
        for i in range(n_layers):
            L = self.layers[i]
            kc, vc = self.k_cache[i, :1], self.vt_cache[i, :1]
            kcl = self.k_cache_lo[i, :1] if sp else None
            vcl = self.vt_cache_lo[i, :1] if sp else None
            a_done, o_done = cnt[2 * i: 2 * i + 1], cnt[2 * i + 1: 2 * i + 2]
            ops.gemm16_rmsnorm_a(h, L.ln1, d.rms_norm_eps, L.wqkv, 3 * H, ops.EPI_F32, sp, c=ws["qkv"])
            ev.record(main)
            side.wait_event(ev)                                  # o_proj becomes resident when q|k|v has finished, i.e. while the attention runs
            ops.attn_decode_rope_chain(ws["qkv"], 1, nh, hd, pos0, self.cos, self.sin, kc, vc, ws["att"], kcl, vcl, ws["att_lo"], a_done)
            with torch.cuda.stream(side):
                ops.gemv16_dma_chain(L.wo, H, ops.EPI_RESID, sp, a_hi=ws["att"], a_lo=ws["att_lo"], c=h, resid=h,
                                     wait=a_done, wait_target=ep * nh, signal=o_done)
            ops.gemv16_dma_chain(L.wgu, 2 * I, ops.EPI_SWIGLU_SPLIT if sp else ops.EPI_SWIGLU16, sp, x=h, norm_w=L.ln2, eps=d.rms_norm_eps,
                                 out_hi=ws["act"], out_lo=ws["act_lo"], wait=o_done, wait_target=ep * ch["blocks_o"])
            ops.gemm16(ws["act"], ws["act_lo"], L.wdown, None, H, ops.EPI_RESID, c=h, resid=h)
        main.wait_stream(side)                                   # (the data dependency is already met through O_i: this is for the allocator's sake)

    def _prefill_layer_rows(self, ws, L, i: int, b0: int, b1: int, s: int, pos0: int) -> None:
        """One decoder layer of a PREFILL on batch rows b0 .. b1 - 1 (the fused-RoPE path of _layers_forward on row slices)."""
        d = self.dims
        H, I, nh, hd = d.hidden_size, d.intermediate_size, d.num_attention_heads, d.head_dim
        sp, nb = self.split, b1 - b0
        r0, r1 = b0 * s, b1 * s
        h = ws["h"][r0:r1]
        x16, att, act, q = ws["x16"][r0:r1], ws["att"][r0:r1], ws["act"][r0:r1], ws["q"][b0:b1]
        x16l = ws["x16_lo"][r0:r1] if sp else None
        attl = ws["att_lo"][r0:r1] if sp else None
        actl = ws["act_lo"][r0:r1] if sp else None
        ql = ws["q_lo"][b0:b1] if sp else None
        kc, vc = self.k_cache[i, b0:b1], self.vt_cache[i, b0:b1]
        kcl = self.k_cache_lo[i, b0:b1] if sp else None
        vcl = self.vt_cache_lo[i, b0:b1] if sp else None
        ops.rmsnorm_bf16(h, L.ln1, d.rms_norm_eps, x16, x16l)
        ops.gemm16_fragw_rope_qkv(x16, x16l, L.wqkv_rope, H, nb, s, nh, pos0, self.cos, self.sin, q, kc, vc, ql, kcl, vcl)
        ops.attn_prefill(q, kc, vc, nb, s, nh, hd, pos0, att, ql, kcl, vcl, attl)
        ops.gemm16(att, attl, L.wo, None, H, ops.EPI_RESID, c=h, resid=h)
        ops.rmsnorm_bf16(h, L.ln2, d.rms_norm_eps, x16, x16l)
        ops.gemm16(x16, x16l, L.wgu, None, 2 * I, ops.EPI_SWIGLU_SPLIT if sp else ops.EPI_SWIGLU16, out_hi=act, out_lo=actl)
        ops.gemm16(act, actl, L.wdown, None, H, ops.EPI_RESID, c=h, resid=h)

    def _prefill_two_streams(self, ws, batch: int, s: int, pos0: int, n_layers: int) -> None:
        """The layer stack of a prefill as two half-batches, each on its own stream, enqueued layer by layer in alternation.
        The side streams work on row slices of `ws`, tensors allocated on the CALLER's stream: safe only because the workspace is
        persistent (owned by the engine, never returned to the caching allocator while a forward is in flight) and both side streams
        are joined into the caller's stream before this returns."""
        if self._side_streams is None:
            self._side_streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        cur = torch.cuda.current_stream()
        half = batch // 2
        parts = ((0, half), (half, batch))
        for st in self._side_streams:
            st.wait_stream(cur)
        for i in range(n_layers):
            L = self.layers[i]
            for st, (b0, b1) in zip(self._side_streams, parts):
                with torch.cuda.stream(st):
                    self._prefill_layer_rows(ws, L, i, b0, b1, s, pos0)
        for st in self._side_streams:
            cur.wait_stream(st)

    def _decode_body(self, st) -> None:
        """One decode step on static buffers: ids -> logits, new K/V written at *pos."""
        d = self.dims
        B = st["ids"].shape[0]
        ws = st["ws"]
        ops.embed_gather(st["ids"].view(-1), self.embed, ws["h"])
        if not self._layers_forward(ws, B, 1, 0, None, pos_dev=st["pos"]):
            ops.rmsnorm_bf16(ws["h"], self.norm, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
        ops.gemm16(ws["x16"], ws["x16_lo"], self.lm_head, None, d.vocab_size, ops.EPI_F32, c=st["logits"])

    def _decode_step_graph(self, input_ids: torch.Tensor, pos0: int) -> torch.Tensor:
        """Greedy-decode step as ONE hipGraph replay (opt-in: `decode_graph` / LLARK_DECODE_GRAPH=1).  Bit-identical
        to the eager launches; measured on MI355X / ROCm 7.2 it does NOT shorten the step (7.41 vs 7.52 ms, B=8): the
        ~260 kernels of a step each keep their ~10 us dispatch + drain boundary inside the graph, so the lever for
        decode is fewer kernels (fusion), not graph replay.  First call per batch size runs eagerly on the static
        buffers (first-launch attribute calls happen outside capture), the second captures, later ones replay."""
        d = self.dims
        B = input_ids.shape[0]
        st = self._dec.get(B)
        if st is None:
            ws = dict(self._workspace(B, 1))                  # private copies: the graph owns these buffers
            ws = {k: (torch.empty_like(v) if v is not None else None) for k, v in ws.items()}
            st = {"ids": torch.zeros((B, 1), dtype=torch.int64, device=self.device),
                  "pos": torch.zeros((1,), dtype=torch.int32, device=self.device),
                  "logits": torch.empty((B, d.vocab_size), dtype=torch.float32, device=self.device),
                  "ws": ws, "graph": None, "calls": 0}
            self._dec[B] = st
        st["ids"].copy_(input_ids)
        st["pos"].fill_(pos0)
        if self.decode_replay and not self.decode_graph:
            if st.get("list") is None:
                with ops.LaunchList.record() as ll:
                    self._decode_body(st)
                st["list"] = ll
            else:
                st["list"].replay()
        elif st["graph"] is not None:
            st["graph"].replay()
        elif st["calls"] == 0:
            self._decode_body(st)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._decode_body(st)
            st["graph"] = g
            g.replay()
        st["calls"] += 1
        self.cur_len = pos0 + 1
        return st["logits"].clone().view(B, 1, d.vocab_size)

    @property
    def fuse_decode_norm_a(self) -> str:
        """"auto" | "1" | "0" (see __init__); assigning a bool means "1" / "0" (it was a bool before round 3)."""
        return self._fuse_decode_norm_a

    @fuse_decode_norm_a.setter
    def fuse_decode_norm_a(self, v) -> None:
        if isinstance(v, bool):
            v = "1" if v else "0"
        v = str(v).strip().lower()
        if v not in ("auto", "1", "0"):
            raise ValueError(f"fuse_decode_norm_a must be 'auto', '1', '0' or a bool, got {v!r}")
        self._fuse_decode_norm_a = v

    def reset(self, batch: int) -> None:
        if self.k_cache is not None and self.k_cache.shape[1] != batch:
            self.k_cache = self.vt_cache = self.k_cache_lo = self.vt_cache_lo = None
        self._ensure_cache(batch)
        self.cur_len, self.cur_batch = 0, batch

    def forward_tokens(self, input_ids: torch.Tensor, audio_segments: Sequence[Tuple[int, int, torch.Tensor]] = (),
                       pos0: int = 0, last_only: bool = False, num_layers: Optional[int] = None,
                       return_hidden: bool = False, hidden_sink: Optional[list] = None) -> torch.Tensor:
        """input_ids (B,S) int64 on device.  audio_segments: (batch index, position of <audio_start>,
        frames fp32 (F, mm) on device): projected frames overwrite rows start+1 .. start+F of the
        embedded sequence (the splice of m2t/models/llamav2.py:141-222).  Returns fp32 logits
        (B,S,V) (or (B,1,V) if last_only).  Positions pos0..pos0+S-1 of the KV cache are written."""
        d = self.dims
        assert self.embed is not None and all(L is not None for L in self.layers[: (num_layers or d.num_hidden_layers)]), \
            "weights not loaded"
        B, S = input_ids.shape
        if pos0 == 0:
            self.reset(B)
        assert B == self.cur_batch and pos0 == self.cur_len, "KV cache is out of sync with the requested positions"
        if pos0 + S > self.smax:
            raise ValueError(f"sequence of {pos0 + S} exceeds the engine's max_seq {self.smax}")
        if (S == 1 and pos0 > 0 and (self.decode_graph or self.decode_replay) and not audio_segments and num_layers is None
                and not return_hidden and hidden_sink is None and (self.decode_replay or not ops.kernel_timing_active())):
            return self._decode_step_graph(input_ids, pos0)
        ws = self._workspace(B, S)
        h = ws["h"]
        ops.embed_gather(input_ids.reshape(-1).contiguous(), self.embed, h)
        for (b, start, frames) in audio_segments:
            assert self.proj_w is not None, "mm_projector weights not loaded"
            F = frames.shape[0]
            a16, a16_lo = ops.split16(frames.contiguous(), torch.bfloat16, want_lo=self.split)
            r0 = b * S + start + 1
            ops.gemm16(a16, a16_lo, self.proj_w, self.proj_b, d.hidden_size, ops.EPI_F32, c=h[r0: r0 + F])
        normed = self._layers_forward(ws, B, S, pos0, num_layers, hidden_sink=hidden_sink)
        self.cur_len = pos0 + S
        if hidden_sink is not None:
            # HF appends model.norm(h) last (LlamaModel.forward): here the kernel's output planes, hi (+ lo) -- 16 significant bits in the
            # "split" flow, the bf16 value in the "bf16" flow
            ops.rmsnorm_bf16(h, self.norm, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
            nh_ = ws["x16"].float() + (ws["x16_lo"].float() if ws["x16_lo"] is not None else 0.0)
            hidden_sink.append(nh_.view(B, S, d.hidden_size))
            normed = True
        if return_hidden:
            return h.view(B, S, d.hidden_size)
        if last_only and S > 1:
            hl = h.view(B, S, d.hidden_size)[:, -1].contiguous()
            x16 = torch.empty((B, d.hidden_size), dtype=torch.bfloat16, device=self.device)
            x16_lo = torch.empty_like(x16) if self.split else None
            ops.rmsnorm_bf16(hl, self.norm, d.rms_norm_eps, x16, x16_lo)
            logits = torch.empty((B, d.vocab_size), dtype=torch.float32, device=self.device)
            ops.gemm16(x16, x16_lo, self.lm_head, None, d.vocab_size, ops.EPI_F32, c=logits)
            return logits.view(B, 1, d.vocab_size)
        logits = torch.empty((B * S, d.vocab_size), dtype=torch.float32, device=self.device)
        if S == 1 and B <= 16 and not normed and (self.fuse_decode_norm_a == "1" or (self.fuse_decode_norm_a == "auto" and ops.gemv_dma_rmsnorm_takes(B, d.vocab_size, d.hidden_size))):
            ops.gemm16_rmsnorm_a(h, self.norm, d.rms_norm_eps, self.lm_head, d.vocab_size, ops.EPI_F32, self.split, c=logits)
            return logits.view(B, S, d.vocab_size)
        if not normed:
            ops.rmsnorm_bf16(h, self.norm, d.rms_norm_eps, ws["x16"], ws["x16_lo"])
        ops.gemm16(ws["x16"], ws["x16_lo"], self.lm_head, None, d.vocab_size, ops.EPI_F32, c=logits)
        return logits.view(B, S, d.vocab_size)
