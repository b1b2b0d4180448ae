"""Training step of the MPT backbone on the HIP kernels (the ``m2t/train.py`` path with ``--model_name_or_path`` an MPT
checkpoint: scripts/training/train_mpt_model.sh -> HF Trainer -> ``WrappedMPTForCausalLM.forward(labels=...)`` m2t/models/mpt.py
:259-330 -> loss.backward() -> AdamW).  Same design as ``HipLlamaTrainer`` (llark_amd/m2t/train_engine.py): forward with
saved activations, full backward through the same MFMA GEMM on transposed operands, one flat fp32 gradient buffer (bucketed
all-reduce), fused AdamW.  What differs from Llama: LayerNorm instead of RMSNorm (``llark_layernorm_bwd``), exact-GELU MLP
(``llark_gelu_bwd``), ALiBi inside the flash-style attention backward (``llark_attn_backward_bf16`` with the slopes),
no rotation (the RoPE kernels run with an identity table), optional biases and ``qk_ln``, tied ``wte`` / ``lm_head``.

``train_wte``: the reference recipe ends with ``wte.requires_grad = False`` (initialize_audio_tokenizer sets the INPUT
embeddings trainable and then freezes the OUTPUT embeddings, which is the same tied tensor, m2t/models/mpt.py:405-411), so
the default keeps ``wte`` frozen; ``train_wte=True`` accumulates both of its uses (gather rows and the logits product).
``clip_qkv`` (clamp in the forward, gradient mask ``llark_clamp_bwd_bf16`` in the backward) and ``logit_scale`` (logits scaled in
place before the loss, the factor folded into the scale of d(logits)) follow ``attn_config`` / ``config.logit_scale``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .. import ops
from .mpt_engine import HipMptEngine

_BF = torch.bfloat16


class HipMptTrainer:
    def __init__(self, engine: HipMptEngine, lr: float = 5e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 train_wte: bool = False, grad_comm: torch.dtype = torch.float32):
        d = engine.dims
        self.grad_comm = grad_comm
        if engine.split:
            raise ValueError("the training step runs in the reference's bf16 flow: build the engine with precision='bf16'")
        self.eng, self.lr, self.betas, self.eps, self.wd = engine, lr, betas, eps, weight_decay
        self.train_wte, self.step_count = train_wte, 0
        for B in engine.blocks:                                  # weights change in place from now on
            for t in (B.wqkv, B.wo, B.wup, B.wdown):
                ops.detach_frag(t)
        ops.detach_frag(engine.wte)
        self.params: List[Tuple[str, torch.Tensor]] = []
        for i, B in enumerate(engine.blocks):
            for nm in ("wqkv", "bqkv", "wo", "bo", "wup", "bup", "wdown", "bdown", "n1w", "n1b", "n2w", "n2b", "qlw", "qlb", "klw", "klb"):
                t = getattr(B, nm)
                if t is not None:
                    self.params.append((f"blocks.{i}.{nm}", t))
        self.params.append(("normf_w", engine.normf_w))
        if engine.normf_b is not None:
            self.params.append(("normf_b", engine.normf_b))
        if train_wte:
            self.params.append(("wte", engine.wte))
        if engine.proj_w is not None:
            self.params += [("proj_w", engine.proj_w), ("proj_b", engine.proj_b)]
        total = sum(p.numel() for _, p in self.params)
        dev = engine.device
        self.flat_grad = torch.zeros((total,), dtype=torch.float32, device=dev)
        self.flat_m, self.flat_v = torch.zeros_like(self.flat_grad), torch.zeros_like(self.flat_grad)
        self.grads: Dict[str, torch.Tensor] = {}
        self._slices: Dict[str, Tuple[int, int]] = {}
        off = 0
        for name, p in self.params:
            self.grads[name] = self.flat_grad[off: off + p.numel()].view(p.shape)
            self._slices[name] = (off, p.numel())
            off += p.numel()

    def zero_grad(self) -> None:
        self.flat_grad.zero_()

    # ---- product helpers (same formulation as HipLlamaTrainer) -----------------------------------
    @staticmethod
    def _dw(dy16: torch.Tensor, x16: torch.Tensor, grad: torch.Tensor) -> None:
        """grad[N][K] += dY^T . X  (both operands contraction-major as they stand: llark_gemm16_t, no transposed copies, when the
        token count is a multiple of 64)"""
        n, k = dy16.shape[1], x16.shape[1]
        if dy16.shape[0] % 64 == 0 and n % 8 == 0 and k % 8 == 0 and dy16.stride(0) % 8 == 0 and x16.stride(0) % 8 == 0:
            ops.gemm16_t(dy16, x16, n, k, dy16.shape[0], True, True, grad, accumulate=True)
            return
        dyT, xT = ops.transposed16(dy16), ops.transposed16(x16)
        ops.gemm16(dyT, None, xT, None, x16.shape[1], ops.EPI_RESID, c=grad, resid=grad, m=dy16.shape[1])

    @staticmethod
    def _dx(dy16: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> None:
        """out[rows][K] = dY . W   (w [N][K]: its row is the contraction index -> llark_gemm16_t on W as stored)"""
        n, k = w.shape
        if n % 64 == 0 and k % 8 == 0 and dy16.shape[1] >= n and dy16.stride(0) % 8 == 0:
            ops.gemm16_t(dy16, w, dy16.shape[0], k, n, False, True, out)
            return
        wT = ops.transposed16(w)
        if dy16.shape[1] < wT.shape[1]:
            pad = torch.zeros((dy16.shape[0], wT.shape[1]), dtype=_BF, device=dy16.device)
            pad[:, : dy16.shape[1]] = dy16
            dy16 = pad
        ops.gemm16(dy16, None, wT, None, w.shape[1], ops.EPI_F32, c=out)

    def _to16(self, x: torch.Tensor, width: int) -> torch.Tensor:
        hi, _ = ops.split16(x, _BF, want_lo=False, kmult=64)
        return hi[:, :width] if hi.shape[1] == width else hi[:, :width].contiguous()

    # ---------------------------------------------------------------------------------------------
    def forward_backward(self, input_ids: torch.Tensor, audio_segments, labels: torch.Tensor, loss_scale: float = 1.0) -> torch.Tensor:
        eng, d, dev = self.eng, self.eng.dims, self.eng.device
        B, S = input_ids.shape
        rows, D, E, nh = B * S, d.d_model, d.expansion_ratio * d.d_model, d.n_heads
        V = d.vocab_size
        f32, bf = dict(dtype=torch.float32, device=dev), dict(dtype=_BF, device=dev)
        eng.reset(B)
        ids_flat = input_ids.reshape(-1).contiguous()
        h = torch.empty((rows, D), **f32)
        ops.embed_gather(ids_flat, eng.wte, h)
        seg_rows, seg_a16 = [], []
        for (b, start, frames) in audio_segments:
            F = frames.shape[0]
            a16, _ = ops.split16(frames.contiguous(), _BF, want_lo=False, kmult=64)
            r0 = b * S + start + 1
            ops.gemm16(a16, None, eng.proj_w, eng.proj_b, D, ops.EPI_F32, c=h[r0: r0 + F])
            seg_rows.append(torch.arange(r0, r0 + F, device=dev))
            seg_a16.append(a16[:, : d.mm_hidden_size])
        # ---------------- forward ----------------
        saved = []
        for i, Bk in enumerate(eng.blocks):
            st = {"h_in": h.clone()}
            x1 = torch.empty((rows, D), **bf)
            ops.layernorm_bf16(h, Bk.n1w, Bk.n1b, d.ln_eps, x1)
            qkv = torch.empty((rows, 3 * D), **f32)
            ops.gemm16(x1, None, Bk.wqkv, Bk.bqkv, 3 * D, ops.EPI_F32, c=qkv)
            if d.clip_qkv:
                st["qkv_raw"] = qkv.clone()              # the clamp's gradient mask needs the values before it
                ops.clamp_f32_(qkv, d.clip_qkv)
            if d.qk_ln:
                st["qkv_pre"] = qkv.clone()
                ops.layernorm_f32_(qkv[:, :D], Bk.qlw, Bk.qlb, d.ln_eps)
                ops.layernorm_f32_(qkv[:, D: 2 * D], Bk.klw, Bk.klb, d.ln_eps)
            q = torch.empty((B, nh, S, 128), **bf)
            kc, vc = eng.k_cache[i], eng.vt_cache[i]
            ops.rope_split_heads(qkv, B, S, nh, 128, 0, eng.cos, eng.sin, q, kc, vc)
            att = torch.empty((rows, D), **bf)
            lse = torch.empty((B * nh, S), **f32)                         # per-query log-sum-exp: the backward recomputes P from it
            ops.attn_prefill_lse(q, kc, vc, B, S, nh, 128, att, lse, alibi_slopes=eng.slopes)
            ops.gemm16(att, None, Bk.wo, Bk.bo, D, ops.EPI_RESID, c=h, resid=h)
            st.update(x1=x1, q=q, att=att, lse=lse, h_mid=h.clone())
            x2 = torch.empty((rows, D), **bf)
            ops.layernorm_bf16(h, Bk.n2w, Bk.n2b, d.ln_eps, x2)
            up = torch.empty((rows, E), **f32)
            ops.gemm16(x2, None, Bk.wup, Bk.bup, E, ops.EPI_F32, c=up)
            act = torch.empty((rows, E), **bf)
            ops.gelu_split_bf16(up, act)
            ops.gemm16(act, None, Bk.wdown, Bk.bdown, D, ops.EPI_RESID, c=h, resid=h)
            st.update(x2=x2, up=up, act=act)
            saved.append(st)
        xf = torch.empty((rows, D), **bf)
        ops.layernorm_bf16(h, eng.normf_w, eng.normf_b, d.ln_eps, xf)
        logits = torch.empty((rows, V), **f32)
        ops.gemm16(xf, None, eng.wte, None, V, ops.EPI_F32, c=logits)
        dlogits = torch.empty((rows, ops.round_up(V, 64)), **bf)
        lscale = 1.0
        if d.logit_scale is not None:                    # modeling_mpt.py:410-416: logits *= logit_scale; d(raw logits) = scale * d(logits)
            lscale = float(d.logit_scale)
            ops.scale_f32_(logits, lscale)
        loss = ops.cross_entropy_fwd_bwd(logits.view(B, S, V), labels.to(dev), dlogits, loss_scale * lscale)
        del logits
        # ---------------- backward ----------------
        g = self.grads
        dtmp = torch.empty((rows, D), **f32)
        self._dx(dlogits, eng.wte, dtmp)
        if self.train_wte:
            self._dw(dlogits[:, :V].contiguous() if dlogits.shape[1] != V else dlogits, xf, g["wte"])
        dh = torch.empty((rows, D), **f32)
        ops.layernorm_bwd(h, eng.normf_w, dtmp, d.ln_eps, dh, g["normf_w"], g.get("normf_b"), False)
        del dlogits
        Sp, BH, scale, smax = ops.round_up(S, 64), B * nh, 1.0 / math.sqrt(128.0), eng.smax
        for i in reversed(range(len(eng.blocks))):
            Bk, st, pre = eng.blocks[i], saved[i], f"blocks.{i}."
            # ---- MLP ----
            dh16 = self._to16(dh, D)
            dact = torch.empty((rows, E), **f32)
            self._dx(dh16, Bk.wdown, dact)
            self._dw(dh16, st["act"], g[pre + "wdown"])
            if Bk.bdown is not None:
                ops.colsum_add(dh, g[pre + "bdown"])
            dup16 = torch.empty((rows, E), **bf)
            dup32 = torch.empty((rows, E), **f32) if Bk.bup is not None else None
            ops.gelu_bwd(st["up"], dact, dup16, dup32)
            del dact
            self._dx(dup16, Bk.wup, dtmp)
            self._dw(dup16, st["x2"], g[pre + "wup"])
            if Bk.bup is not None:
                ops.colsum_add(dup32, g[pre + "bup"])
            del dup16, dup32
            ops.layernorm_bwd(st["h_mid"], Bk.n2w, dtmp, d.ln_eps, dh, g[pre + "n2w"], g.get(pre + "n2b"), True)
            # ---- attention ----
            dh16 = self._to16(dh, D)
            self._dx(dh16, Bk.wo, dtmp)                                    # d(context)
            self._dw(dh16, st["att"], g[pre + "wo"])
            if Bk.bo is not None:
                ops.colsum_add(dh, g[pre + "bo"])
            dctx16 = self._to16(dtmp, D)
            dO = torch.empty((BH, S, 128), **bf)
            ops.split_heads16(dctx16, B, S, nh, 128, dO)
            q = st["q"].view(BH, S, 128)
            kc, vtc = eng.k_cache[i], eng.vt_cache[i]
            # flash-style backward (csrc/attn_bwd.hip), ALiBi inside: P is recomputed per tile from the forward's log-sum-exp
            v_rm = torch.empty((BH, S, 128), **bf)
            ops.transpose16(vtc, smax, 128, S, v_rm, 128, BH, 128 * smax, S * 128)
            dq, dk, dv = (torch.empty((BH, S, 128), **f32) for _ in range(3))
            dsum = torch.empty((BH, S), **f32)
            ops.attn_backward(q, kc, v_rm, dO, st["att"], st["lse"], dsum, B, S, nh, 128, dq, dk, dv,
                              alibi_slopes=eng.slopes)
            dqkv = torch.empty((rows, 3 * D), **bf)
            ops.rope_merge_bwd(dq, dk, dv, eng.cos, eng.sin, B, S, nh, 128, 0, dqkv)      # identity rotation: heads -> [rows][3D]
            if d.qk_ln:                                    # through the LayerNorms over q and k (fp32), then back to bf16
                d32 = dqkv.float()
                dpre = torch.empty((rows, 2 * D), **f32)
                ops.layernorm_bwd(st["qkv_pre"][:, :D], Bk.qlw, d32[:, :D], d.ln_eps, dpre[:, :D], g[pre + "qlw"], g.get(pre + "qlb"), False)
                ops.layernorm_bwd(st["qkv_pre"][:, D: 2 * D], Bk.klw, d32[:, D: 2 * D], d.ln_eps, dpre[:, D:], g[pre + "klw"], g.get(pre + "klb"), False)
                d32[:, : 2 * D] = dpre
                dqkv = self._to16(d32, 3 * D)
                del d32, dpre
            if d.clip_qkv:
                ops.clamp_bwd_bf16_(st["qkv_raw"], d.clip_qkv, dqkv)
            self._dx(dqkv, Bk.wqkv, dtmp)
            self._dw(dqkv, st["x1"], g[pre + "wqkv"])
            if Bk.bqkv is not None:
                ops.colsum_add(dqkv.float(), g[pre + "bqkv"])
            ops.layernorm_bwd(st["h_in"], Bk.n1w, dtmp, d.ln_eps, dh, g[pre + "n1w"], g.get(pre + "n1b"), True)
            saved[i] = None
        # ---- bottom: projector, embedding rows ----
        if seg_rows:
            ridx = torch.cat(seg_rows)
            dya = torch.empty((ridx.numel(), D), **f32)
            ops.gather_rows(dh, ridx, dya)
            ops.colsum_add(dya, g["proj_b"])
            self._dw(self._to16(dya, D), torch.cat(seg_a16, dim=0).contiguous(), g["proj_w"])
        if self.train_wte:
            keep = torch.ones((rows,), dtype=torch.bool, device=dev)
            if seg_rows:
                keep[torch.cat(seg_rows)] = False
            ridx = keep.nonzero().reshape(-1)
            tmp = torch.empty((ridx.numel(), D), **f32)
            ops.gather_rows(dh, ridx, tmp)
            ops.scatter_add_rows(tmp, ids_flat[ridx].contiguous(), g["wte"])
        return loss

    def allreduce_grads(self, world: int, bucket_elems: int = 64 * 1024 * 1024) -> None:
        if world <= 1:
            return
        from .. import dist as D

        n = self.flat_grad.numel()
        comm = getattr(self, "grad_comm", torch.float32)
        for wk in [D.all_reduce_sum_async(self.flat_grad[o: min(o + bucket_elems, n)], comm) for o in range(0, n, bucket_elems)]:
            wk.wait()

    def step(self, world: int = 1, max_grad_norm=None) -> None:
        """AdamW; ``max_grad_norm`` = HF Trainer's clip_grad_norm_, coefficient formed on the device (see HipLlamaTrainer.step)."""
        clip = max_grad_norm is not None and max_grad_norm > 0
        sumsq = ops.sumsq_f32(self.flat_grad) if clip else None
        self.last_grad_sumsq = sumsq
        self.step_count += 1
        b1, b2 = self.betas
        for name, p in self.params:
            off, n = self._slices[name]
            ops.adamw(p.view(-1), self.flat_grad[off: off + n], self.flat_m[off: off + n], self.flat_v[off: off + n], self.lr, b1, b2,
                      self.eps, 0.0 if p.dim() == 1 else self.wd, self.step_count, 1.0 / world,
                      grad_sumsq=sumsq, max_grad_norm=float(max_grad_norm) if clip else 0.0)
        self.zero_grad()

    def export_grads_ref(self) -> Dict[str, torch.Tensor]:
        """Gradients under the reference's state-dict names."""
        names = {"wqkv": "attn.Wqkv.weight", "bqkv": "attn.Wqkv.bias", "wo": "attn.out_proj.weight", "bo": "attn.out_proj.bias",
                 "wup": "ffn.up_proj.weight", "bup": "ffn.up_proj.bias", "wdown": "ffn.down_proj.weight", "bdown": "ffn.down_proj.bias",
                 "n1w": "norm_1.weight", "n1b": "norm_1.bias", "n2w": "norm_2.weight", "n2b": "norm_2.bias",
                 "qlw": "attn.q_ln.weight", "qlb": "attn.q_ln.bias", "klw": "attn.k_ln.weight", "klb": "attn.k_ln.bias"}
        out = {}
        for name, _ in self.params:
            if name.startswith("blocks."):
                _, i, nm = name.split(".")
                out[f"transformer.blocks.{i}.{names[nm]}"] = self.grads[name]
        out["transformer.norm_f.weight"] = self.grads["normf_w"]
        if "normf_b" in self.grads:
            out["transformer.norm_f.bias"] = self.grads["normf_b"]
        if "wte" in self.grads:
            out["transformer.wte.weight"] = self.grads["wte"]
        if "proj_w" in self.grads:
            out["transformer.mm_projector.weight"], out["transformer.mm_projector.bias"] = self.grads["proj_w"], self.grads["proj_b"]
        return out
