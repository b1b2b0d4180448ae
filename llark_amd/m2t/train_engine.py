"""Training step of the LLM half on the HIP kernels: forward with saved activations, full backward, AdamW, and the
data-parallel gradient all-reduce (RCCL over xGMI via ``torch.distributed``; gloo in the CPU tests).

Replaces what the reference reaches through ``trainer.train()`` (m2t/train.py:255-260 -> HF ``Trainer.training_step``
-> ``WrappedLlamav2ForCausalLM.forward(labels=...)`` m2t/models/llamav2.py:259-337, ``loss.backward()``, DDP gradient
all-reduce, ``torch.optim.AdamW``; hyper-parameters scripts/training/train_llark.sh:20-49).

Semantics kept from the reference's default recipe (``--freeze_backbone False --tune_mm_mlp_adapter True --bf16 True``):
  * trainable: every Llama weight, ``mm_projector`` and -- through ``orig_embeds_params`` (m2t/models/llamav2.py:176-198)
    -- only the ``<audio_start>`` / ``<audio_end>`` rows of ``embed_tokens``; ``lm_head`` is frozen
    (m2t/models/llamav2.py:412-415);
  * bf16 parameters and matrix operands, fp32 accumulation; loss = mean shifted CE over labels != -100;
  * gradient accumulation over micro-batches (``--gradient_accumulation_steps``) then one AdamW step.
Differences (documented in DESIGN.md): AdamW moments are fp32 (torch keeps them in the parameter dtype); no
gradient checkpointing (288 GB of HBM hold the ~13 GB of saved activations of a 4 x 512 micro-batch).

Gradients live in ONE flat fp32 buffer (views per tensor, kernel layouts: fused q|k|v rows, interleaved gate/up) so the
data-parallel exchange is a handful of large all-reduces.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops
from .engine import HipLlamaEngine

_BF = torch.bfloat16


class HipLlamaTrainer:
    def __init__(self, engine: HipLlamaEngine, lr: float = 5e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, embed_grad_tokens: Sequence[int] = (), train_embed_all: bool = False,
                 grad_comm: torch.dtype = torch.float32, optimizer_state: bool = True, gradient_checkpointing: bool = False):
        if grad_comm not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_comm must be torch.float32 or torch.bfloat16")
        self.grad_comm = grad_comm                   # transport dtype of the gradient all-reduce (the reference's DDP sends bf16)
        # train_llark.sh:25 `--gradient_checkpointing True` (HF: one checkpoint per decoder layer): the forward keeps only each
        # layer's input; the backward re-runs that layer's forward -- same kernels, same order, hence bit-identical gradients --
        # right before differentiating it.  Activation memory drops from ~0.18 MB to 16 KiB per token and layer at 7B widths.
        self.gradient_checkpointing = bool(gradient_checkpointing)
        if engine.split:
            raise ValueError("the training step runs in the reference's bf16 flow: build the engine with precision='bf16'")
        self.eng = engine
        engine.drop_frag_weights()                    # weights change in place from now on
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.embed_grad_tokens = set(int(t) for t in embed_grad_tokens)
        self.train_embed_all = train_embed_all
        self.step_count = 0
        d, dev = engine.dims, engine.device
        # ---- parameter table: (name, tensor) in kernel layout; lm_head is frozen like the reference ----
        self.params: List[Tuple[str, torch.Tensor]] = []
        for i, L in enumerate(engine.layers):
            for nm in ("wqkv", "wo", "wgu", "wdown", "ln1", "ln2"):
                self.params.append((f"layers.{i}.{nm}", getattr(L, nm)))
        self.params += [("norm", engine.norm), ("embed", engine.embed)]
        if engine.proj_w is not None:
            self.params += [("proj_w", engine.proj_w), ("proj_b", engine.proj_b)]
        self._pname = {p.data_ptr(): n for n, p in self.params}
        total = sum(p.numel() for _, p in self.params)
        self.flat_grad = torch.zeros((total,), dtype=torch.float32, device=dev)
        # AdamW moments (2 x 27 GB at 7B): not allocated on the autograd-bridge path, where a torch optimizer owns the state
        self.flat_m = torch.zeros_like(self.flat_grad) if optimizer_state else None
        self.flat_v = torch.zeros_like(self.flat_grad) if optimizer_state else None
        self.grads: Dict[str, torch.Tensor] = {}
        self._slices: Dict[str, Tuple[int, int]] = {}
        off = 0
        for name, p in self.params:
            n = p.numel()
            self.grads[name] = self.flat_grad[off: off + n].view(p.shape)
            self._slices[name] = (off, n)
            off += n
        self.micro_batches = 0
        self._matrix_grads = {n for n, p in self.params if p.dim() == 2 and n != "embed"}
        self._fresh = set()                            # flat_grad starts zeroed: accumulate until the first zero_grad()
        # Operands DERIVED from the weights, valid until the next optimizer step (self._wver).  The first micro-batch of a step
        # needs none.  From the second micro-batch on, when there is something to amortise over, a fragment-major twin of W is
        # built once so that the forward products take the B-direct kernel like the inference engine (+8 .. 13 %); dX = dY . W
        # runs on W as stored (llark_gemm16_t) unless dx_direct_uses says otherwise.  Only when this object owns the optimizer:
        # on the autograd-bridge path the weights are rebuilt outside (sync_engine).
        self.derived_operands = bool(optimizer_state)
        self._wver = 0
        self._derived: Dict[int, list] = {}
        self._fwd_uses: Dict[int, list] = {}
        self._dx_uses: Dict[int, list] = {}
        # how many dX products per weight and optimizer step read W as it stands (llark_gemm16_t) before a K-contiguous W^T and
        # its fragment-major twin are built for the rest.  Default: all of them -- with the 128 x 256 tile llark_gemm16_t is within
        # 3 % of the B-direct kernel on these shapes, and the twins cost a transpose + a pack of all 13.5 GB of weights per step
        # (25 ms and 24 GB of HBM at 7B: 896 -> 868 ms for 4 micro-batches of 2 x 2048); they would pay from ~16 micro-batches.
        self.dx_direct_uses = int(os.environ.get("LLARK_TRAIN_DX_DIRECT_USES", str(1 << 30)))
        # Round 6: the optimizer itself keeps the operand twins current.  Every Llama weight matrix the twin kernel
        # takes carries TWO persistent fragment-major copies -- W (forward: the B-direct DMA loop from the FIRST micro-batch on) and W^T
        # (dX = dY . W on the same kernel instead of llark_gemm16_t) -- built once here and rewritten by llark_adamw_twins inside the
        # AdamW pass that already reads and writes every parameter (+ 4 B of stores on 22 B per parameter; no transpose16 / pack_frag in
        # the step).  +2 x 13.5 GB at 7B.  The q|k|v twin is stored in the fused-RoPE row order: the training forward takes
        # llark_gemm16_fragw_rope_qkv (RoPE, head split and both cache writes in the epilogue) wherever the shape qualifies.
        self.twins: Dict[str, Tuple[torch.Tensor, torch.Tensor, int]] = {}
        self._frozen_wT: Dict[int, torch.Tensor] = {}
        self.use_twins = bool(optimizer_state) and os.environ.get("LLARK_TRAIN_TWINS", "1") != "0" and os.environ.get("LLARK_FRAG", "1") != "0"
        self.dw_fragw = os.environ.get("LLARK_TRAIN_DW_FRAGW", "1") != "0" and os.environ.get("LLARK_FRAG", "1") != "0"
        # the dW product on the 16x16x32 MFMA shape (csrc/gemm_bda16.hip): more flops per joule under the power limit
        self.dw_mfma16 = self.dw_fragw and os.environ.get("LLARK_TRAIN_DW_MFMA16", "0") != "0"
        self.attn_glue_fused = os.environ.get("LLARK_TRAIN_ATTN_GLUE_FUSED", "1") != "0"
        self.swiglu_fused = self.use_twins and os.environ.get("LLARK_TRAIN_SWIGLU_FUSED", "1") != "0"
        self.rope_fused = self.use_twins and d.head_dim == 128 and d.num_attention_heads % 2 == 0 and os.environ.get("LLARK_TRAIN_ROPE_FUSED", "1") != "0"
        if self.use_twins:
            self._build_twins()
        # gradient-norm bookkeeping (HF Trainer max_grad_norm): sum of squares collected by the dW epilogues of a step's last
        # micro-batch, the flat-gradient spans it already covers, and the total of the last clipped step
        self._norm_collect = False
        self._norm_acc: Optional[torch.Tensor] = None
        self._norm_spans: List[Tuple[int, int]] = []
        self._last_sumsq: Optional[torch.Tensor] = None
        self._exchange_events: list = []
        self.time_phases = False                  # True: forward_backward leaves (start, forward done, backward done) HIP events in phase_events
        self.phase_events: Optional[list] = None

    # ------------------------------------------------------------------------------------------
    def zero_grad(self) -> None:
        """Weight-matrix gradients are WRITTEN by the first micro-batch's dW product (no 27 GB memset, no residual read
        in that epilogue); only the small accumulate-only slices (norm gains, embedding rows, projector bias) are zeroed."""
        for name, _ in self.params:
            if name not in self._matrix_grads:
                self.grads[name].zero_()
        self._fresh = set(self._matrix_grads)
        self.micro_batches = 0

    def _build_twins(self) -> None:
        """Initial construction of the operand twins (afterwards llark_adamw_twins rewrites them with every optimizer step)."""
        eng, d = self.eng, self.eng.dims
        H = d.hidden_size
        order = None
        for i, L in enumerate(eng.layers):
            for nm in ("wqkv", "wo", "wgu", "wdown"):
                w = getattr(L, nm)
                n, k = w.shape
                if not ops.adamw_twins_takes(n, k) or not w.is_contiguous():
                    continue
                rope_rows = 2 * H if (nm == "wqkv" and self.rope_fused) else 0
                if rope_rows:
                    if order is None:
                        order = ops.rope_qkv_row_order(d.num_attention_heads, d.head_dim).to(w.device)
                    wfrag = ops.pack_weight16_frag(w.index_select(0, order), n)
                else:
                    wfrag = ops.pack_weight16_frag(w, n)
                    w._llark_frag = (wfrag, n, k)                 # ops.gemm16 takes the B-direct kernel for >= FRAG_MIN_ROWS rows
                wT = ops.transposed16(w)                          # [k][n] (n % 64 == 0: no padding)
                wtfrag = ops.pack_weight16_frag(wT, k)
                del wT
                self.twins[f"layers.{i}.{nm}"] = (wfrag, wtfrag, rope_rows)
        # lm_head is frozen (m2t/models/llamav2.py:412-415): its twins are built once and never go stale
        lm = eng.lm_head
        if lm is not None and lm.shape[1] % 64 == 0:
            ops.attach_frag(lm, lm.shape[0])
            wT = ops.transposed16(lm)                             # [H][V padded to 64]
            ops.attach_frag(wT, wT.shape[0])
            self._frozen_wT[lm.data_ptr()] = wT

    def _twin_of(self, w: torch.Tensor):
        name = self._pname.get(w.data_ptr())
        return self.twins.get(name) if name is not None else None

    def _dw(self, dy16: torch.Tensor, x16: torch.Tensor, grad: torch.Tensor, name: str) -> None:
        """grad[N][K] (+)= dY^T . X   (dy16 [rows][N], x16 [rows][K] bf16); plain write on the first micro-batch.
        Both operands are contraction-major as they stand (the token index is their row): ``llark_gemm16_t`` reads them through
        the transposing LDS load, no dY^T / X^T copies; token counts that are not a multiple of 64 take the transposing path.
        On the last micro-batch of a step (``_norm_collect``) the product's epilogue also adds the sum of squares of the gradient
        it completes to ``_norm_acc`` (gradient-norm clipping without re-reading the gradients)."""
        sumsq = self._norm_acc if self._norm_collect else None
        n, k = dy16.shape[1], x16.shape[1]
        fresh = name in self._fresh
        self._fresh.discard(name)
        toks = dy16.shape[0]
        if (self.dw_fragw and toks >= ops.FRAG_MIN_ROWS and x16.shape[0] == toks and k % 8 == 0 and x16.stride(0) % 8 == 0
                and ops.gemm16_ta_fragw_takes(n, toks, dy16.stride(0))):
            # round 6: dY as it stands through the DMA loop's transposing LDS read, X^T fragment-major (one transposing pack of X: 2 x its
            # bytes) -- the B operand streams L2 -> VGPR and never waits on a barrier (csrc/gemm_bda.hip: gemm_bda_ta_kernel)
            c16 = self.dw_mfma16 and ops.gemm16_ta_fragw16_takes(n, k, toks, dy16.stride(0), grad)
            xt = ops.pack_frag_t16(x16, k, chunk16=c16)
            ops.gemm16_ta_fragw(dy16, xt, n, k, toks, grad, accumulate=not fresh, sumsq=sumsq, chunk16=c16)
            if sumsq is not None:
                off, cnt = self._slices[name]
                self._norm_spans.append((off, off + cnt))
            return
        if dy16.shape[0] % 64 == 0 and n % 8 == 0 and k % 8 == 0 and dy16.stride(0) % 8 == 0 and x16.stride(0) % 8 == 0:
            ops.gemm16_t(dy16, x16, n, k, dy16.shape[0], True, True, grad, accumulate=not fresh, sumsq=sumsq)
            if sumsq is not None:                          # this gradient's share of the squared norm is in the accumulator already
                off, cnt = self._slices[name]
                self._norm_spans.append((off, off + cnt))
            return
        dyT = ops.transposed16(dy16)
        xT = ops.transposed16(x16)
        if fresh:
            ops.gemm16(dyT, None, xT, None, k, ops.EPI_F32, c=grad, m=n)
        else:
            ops.gemm16(dyT, None, xT, None, k, ops.EPI_RESID, c=grad, resid=grad, m=n)

    def _w_transposed(self, w: torch.Tensor) -> torch.Tensor:
        """The K-contiguous W^T of a dX product that does not go through llark_gemm16_t (shapes it does not take, or -- with
        ``derived_operands`` -- the second and later micro-batches of an optimizer step, which share one W^T with a fragment-major
        twin for the B-direct kernel)."""
        if not self.derived_operands:
            return ops.transposed16(w)
        ent = self._derived.get(w.data_ptr())
        if ent is None or ent[0] != self._wver:
            ent = [self._wver, ops.transposed16(w)]
            self._derived[w.data_ptr()] = ent
            if ent[1].shape[1] % 64 == 0:
                ops.attach_frag(ent[1], ent[1].shape[0])
        return ent[1]

    def _fwd_weight(self, w: torch.Tensor) -> torch.Tensor:
        """A weight about to be multiplied in a forward product: from its second use since the last optimizer step it carries a
        fragment-major twin (ops.gemm16 then takes the B-direct kernel)."""
        if self.twins and self._twin_of(w) is not None:
            return w
        if self.derived_operands:
            ent = self._fwd_uses.get(w.data_ptr())
            if ent is None or ent[0] != self._wver:
                ent = [self._wver, 0]
                self._fwd_uses[w.data_ptr()] = ent
            ent[1] += 1
            if ent[1] == 2 and w.shape[1] % 64 == 0:
                ops.attach_frag(w, w.shape[0])
        return w

    def _weights_changed(self) -> None:
        """After an optimizer step: every derived operand is stale."""
        self._wver += 1
        if self.derived_operands:
            for n, p in self.params:
                if p.dim() == 2 and n not in self.twins:
                    ops.detach_frag(p)

    def _dx(self, dy16: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> None:
        """out[rows][K] = dY . W   (w [N][K] bf16 kernel layout: its row IS the contraction index).
        ``llark_gemm16_t`` on W as it stands (no W^T) for the first ``dx_direct_uses`` products of a weight since the last optimizer
        step -- by default all of them; past that the K-contiguous transpose + fragment-major twin are built once and the B-direct
        kernel (~3 % faster per product) amortises them over the remaining micro-batches."""
        n, k = w.shape
        rows = dy16.shape[0]
        if rows >= ops.FRAG_MIN_ROWS and dy16.stride(0) % 8 == 0:
            tw = self._twin_of(w)
            if tw is not None and dy16.shape[1] >= n:                # W^T fragment-major, kept current by llark_adamw_twins
                ops.gemm16_fragw(dy16, None, tw[1], None, k, n, ops.EPI_F32, c=out)
                return
            wT = self._frozen_wT.get(w.data_ptr())
            if wT is not None and dy16.shape[1] >= wT.shape[1]:
                ops.gemm16(dy16, None, wT, None, k, ops.EPI_F32, c=out)
                return
        if self.derived_operands:
            ent = self._dx_uses.get(w.data_ptr())
            if ent is None or ent[0] != self._wver:
                ent = [self._wver, 0]
                self._dx_uses[w.data_ptr()] = ent
            ent[1] += 1
            first = ent[1] <= self.dx_direct_uses
        else:
            first = True
        if first and n % 64 == 0 and k % 8 == 0 and dy16.shape[1] >= n and dy16.stride(0) % 8 == 0:
            ops.gemm16_t(dy16, w, dy16.shape[0], k, n, False, True, out)
            return
        wT = self._w_transposed(w)                    # [K][Np]
        if dy16.shape[1] < wT.shape[1]:               # pad dY columns up to the 64-multiple K of this product
            pad = torch.zeros((dy16.shape[0], wT.shape[1]), dtype=_BF, device=dy16.device)
            pad[:, : dy16.shape[1]] = dy16
            dy16 = pad
        ops.gemm16(dy16, None, wT, None, w.shape[1], ops.EPI_F32, c=out)

    # ------------------------------------------------------------------------------------------
    def forward_backward(self, input_ids: torch.Tensor, audio_segments, labels: torch.Tensor, loss_scale: float = 1.0,
                         overlap_allreduce_world: int = 1, last_micro_batch: bool = False, loss_groups: int = 1) -> torch.Tensor:
        """One micro-batch: returns the (unscaled) loss as a device scalar and ACCUMULATES gradients.
        ``loss_scale`` = 1 / gradient_accumulation_steps like HF Trainer.
        ``loss_groups`` > 1: the batch holds that many of the recipe's micro-batches (consecutive, equal-sized groups of sequences) run
        as ONE pass -- the accumulation of train_llark.sh:26-27 (per_device_train_batch_size 2 x gradient_accumulation_steps 4) fused:
        the loss is normalised per group (each group's mean over ITS label tokens, times ``loss_scale``), so the gradient is the sum the
        separate micro-batches would accumulate; every dW product then writes its gradient once instead of read-modify-writing 27 GB of
        fp32 per micro-batch.  Returns the mean of the group losses.
        ``last_micro_batch``: this call completes the gradients of an optimizer step: every dW product then also adds the sum of
        squares of the gradient it writes to a device scalar (epilogue of ``llark_gemm16_t_sumsq``) -- ``step(max_grad_norm=...)``
        only reduces what is left (norm gains, embedding rows) instead of re-reading all 27 GB of gradients.  (Single rank only: with
        an all-reduce pending the norm is taken after it, in one pass.)
        ``overlap_allreduce_world`` > 1 (pass it on the LAST micro-batch of an optimizer step, the reference's DDP
        ``no_sync`` boundary): as soon as the backward of decoder layer i is complete its ~0.8 GB gradient slice is
        all-reduced asynchronously (RCCL stream) while layers i-1 ... 0 are still being differentiated; call
        :meth:`allreduce_grads` afterwards -- it only exchanges what is left and waits."""
        eng, d = self.eng, self.eng.dims
        dev = eng.device
        B, S = input_ids.shape
        # last micro-batch on one rank: every dW product also leaves the sum of squares of the gradient it completes (epilogue of
        # llark_gemm16_t_sumsq), so step(max_grad_norm=...) only has to reduce the small rest instead of re-reading 27 GB
        self._norm_collect = bool(last_micro_batch and self.flat_m is not None and overlap_allreduce_world <= 1)
        if self._norm_collect:
            if self._norm_acc is None:
                self._norm_acc = torch.zeros((1,), dtype=torch.float64, device=self.flat_grad.device)
            self._norm_acc.zero_()
        # a collecting call starts from scratch; any OTHER call adds to gradients whose squares an earlier collecting call may
        # have summed: those partial sums are stale either way
        self._norm_spans = []
        ev = None
        if getattr(self, "time_phases", False) and self.flat_grad.is_cuda:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]        # start | forward done | backward done
            ev[0].record()
        rows = B * S
        H, I, nh, hd, V = d.hidden_size, d.intermediate_size, d.num_attention_heads, d.head_dim, d.vocab_size
        f32 = dict(dtype=torch.float32, device=dev)
        bf = dict(dtype=_BF, device=dev)
        eng.reset(B)
        ids_flat = input_ids.reshape(-1).contiguous()
        h = torch.empty((rows, H), **f32)
        ops.embed_gather(ids_flat, eng.embed, h)
        seg_rows, seg_a16 = [], []
        for (b, start, frames) in audio_segments:
            F = frames.shape[0]
            a16, _ = ops.split16(frames.contiguous(), _BF, want_lo=False, kmult=64)
            r0 = b * S + start + 1
            ops.gemm16(a16, None, eng.proj_w, eng.proj_b, H, ops.EPI_F32, c=h[r0: r0 + F])
            seg_rows.append(torch.arange(r0, r0 + F, device=dev))
            seg_a16.append(a16[:, : d.mm_hidden_size])
        fused_rows = rows >= ops.FRAG_MIN_ROWS                          # the fragment-major (B-direct) kernels take the products
        # ---------------- forward, saving what the backward needs ----------------
        def layer_forward(i, L, h):
            """One decoder layer on the residual stream ``h`` (updated in place); returns everything its backward reads."""
            st = {"h_in": h}                                       # the stream is never updated in place: each residual product writes a new buffer
            x1 = torch.empty((rows, H), **bf)
            ops.rmsnorm_bf16(h, L.ln1, d.rms_norm_eps, x1)
            q = torch.empty((B, nh, S, hd), **bf)
            kc, vc = eng.k_cache[i, :B], eng.vt_cache[i, :B]
            tw = self.twins.get(f"layers.{i}.wqkv")
            v_rm = None
            if tw is not None and tw[2] and S >= 32 and fused_rows:     # RoPE / head split / K and V^T writes in the product's epilogue
                if self.attn_glue_fused:                                 # ... and V row-major for the backward (no transpose16 of the V^T cache)
                    v_rm = torch.empty((B * nh, S, hd), **bf)
                    ops.gemm16_fragw_rope_qkv_train(x1, tw[0], H, B, S, nh, 0, eng.cos, eng.sin, q, kc, vc, v_rm)
                else:
                    ops.gemm16_fragw_rope_qkv(x1, None, tw[0], H, B, S, nh, 0, eng.cos, eng.sin, q, kc, vc)
            else:
                qkv = torch.empty((rows, 3 * H), **f32)
                ops.gemm16(x1, None, self._fwd_weight(L.wqkv), None, 3 * H, ops.EPI_F32, c=qkv)
                ops.rope_split_heads(qkv, B, S, nh, hd, 0, eng.cos, eng.sin, q, kc, vc)
                del qkv
            att = torch.empty((rows, H), **bf)
            lse = torch.empty((B * nh, S), **f32)                         # per-query log-sum-exp: the backward recomputes P from it
            ops.attn_prefill_lse(q, kc, vc, B, S, nh, hd, att, lse)
            h_mid = torch.empty_like(h)
            ops.gemm16(att, None, self._fwd_weight(L.wo), None, H, ops.EPI_RESID, c=h_mid, resid=h)
            h = h_mid
            st.update(x1=x1, q=q, att=att, lse=lse, h_mid=h_mid, v_rm=v_rm)
            x2 = torch.empty((rows, H), **bf)
            ops.rmsnorm_bf16(h, L.ln2, d.rms_norm_eps, x2)
            act = torch.empty((rows, I), **bf)
            tw = self.twins.get(f"layers.{i}.wgu")
            gu = None
            if tw is not None and fused_rows and self.swiglu_fused and self.twins.get(f"layers.{i}.wdown") is not None:
                gu = torch.empty((rows, 2 * I), **bf)                    # SwiGLU in the epilogue; gate | up kept as bf16 for the backward
                if not ops.gemm16_fragw_swiglu_train(0, x2, tw[0], 2 * I, H, act, gu):
                    gu = None
            if gu is None:
                gu = torch.empty((rows, 2 * I), **f32)
                ops.gemm16(x2, None, self._fwd_weight(L.wgu), None, 2 * I, ops.EPI_F32, c=gu)
                ops.swiglu_fwd(gu, act)
            h_out = torch.empty_like(h)
            ops.gemm16(act, None, self._fwd_weight(L.wdown), None, H, ops.EPI_RESID, c=h_out, resid=h)
            st.update(x2=x2, gu=gu, act=act, h_out=h_out)
            return st

        saved = []
        for i, L in enumerate(eng.layers):
            st = layer_forward(i, L, h)
            h = st.pop("h_out")
            saved.append({"h_in": st["h_in"]} if self.gradient_checkpointing else st)
            del st
        xf = torch.empty((rows, H), **bf)
        ops.rmsnorm_bf16(h, eng.norm, d.rms_norm_eps, xf)
        logits = torch.empty((rows, V), **f32)
        ops.gemm16(xf, None, eng.lm_head, None, V, ops.EPI_F32, c=logits)
        Vp = ops.round_up(V, 64)
        dlogits = torch.empty((rows, Vp), **bf)
        if loss_groups > 1:
            assert B % loss_groups == 0, "loss_groups must divide the batch"
            gb, lab = B // loss_groups, labels.to(dev)
            lg = logits.view(B, S, V)
            loss = None
            for gi in range(loss_groups):
                li = ops.cross_entropy_fwd_bwd(lg[gi * gb: (gi + 1) * gb], lab[gi * gb: (gi + 1) * gb], dlogits[gi * gb * S: (gi + 1) * gb * S], loss_scale)
                loss = li if loss is None else loss + li
            loss = loss / loss_groups
        else:
            loss = ops.cross_entropy_fwd_bwd(logits.view(B, S, V), labels.to(dev), dlogits, loss_scale)
        del logits
        # ---------------- backward ----------------
        if ev is not None:
            ev[1].record()
        g = self.grads
        dtmp = torch.empty((rows, H), **f32)
        self._dx(dlogits, eng.lm_head, dtmp)                           # d(norm output); lm_head itself is frozen
        dh = torch.empty((rows, H), **f32)
        # round 6: every RMSNorm backward also leaves the bf16 copy of its dx (the next products' A operand): no split16 pass over dh
        fuse16 = H % 64 == 0 and os.environ.get("LLARK_TRAIN_NORM_BWD_OUT16", "1") != "0"
        dh16 = torch.empty((rows, H), **bf) if fuse16 else None
        ops.rmsnorm_bwd(h, eng.norm, dtmp, d.rms_norm_eps, dh, False, g["norm"], dx16=dh16)
        del dlogits
        Sp = ops.round_up(S, 64)
        BH = B * nh
        scale = 1.0 / math.sqrt(hd)
        smax = eng.smax
        for i in reversed(range(len(eng.layers))):
            L, st = eng.layers[i], saved[i]
            if self.gradient_checkpointing:          # recompute this layer's forward from its saved input (K / V caches included)
                st = layer_forward(i, L, st["h_in"])
                st.pop("h_out")
                saved[i] = None
            pre = f"layers.{i}."
            # ---- MLP ----
            if not fuse16:
                dh16, _ = ops.split16(dh, _BF, want_lo=False, kmult=64)
                dh16 = dh16[:, :H]
            dgu = torch.empty((rows, 2 * I), **bf)
            tw = self.twins.get(pre + "wdown")
            if not (st["gu"].dtype == _BF and tw is not None and
                    ops.gemm16_fragw_swiglu_train(1, dh16, tw[1], I, H, dgu, st["gu"])):   # d(act) never leaves the accumulators
                dact = torch.empty((rows, I), **f32)
                self._dx(dh16, L.wdown, dact)
                if st["gu"].dtype == _BF:
                    raise RuntimeError("train_engine: bf16 gate|up was saved but the fused SwiGLU backward declined the shape")
                ops.swiglu_bwd(st["gu"], dact, dgu)
                del dact
            self._dw(dh16, st["act"], g[pre + "wdown"], pre + "wdown")
            self._dx(dgu, L.wgu, dtmp)
            self._dw(dgu, st["x2"], g[pre + "wgu"], pre + "wgu")
            del dgu
            ops.rmsnorm_bwd(st["h_mid"], L.ln2, dtmp, d.rms_norm_eps, dh, True, g[pre + "ln2"], dx16=dh16)
            # ---- attention ----
            if not fuse16:
                dh16, _ = ops.split16(dh, _BF, want_lo=False, kmult=64)
                dh16 = dh16[:, :H]
            q = st["q"].view(BH, S, hd)
            kc = eng.k_cache[i, :B]                                      # [B][nh][smax][hd]
            vtc = eng.vt_cache[i, :B]                                    # [B][nh][hd][smax]
            # flash-style backward (csrc/attn_bwd.hip): P is recomputed per tile from the forward's log-sum-exp, no S x S matrix
            # exists and no operand is transposed
            v_rm = st.get("v_rm")
            if v_rm is None:                                             # (V back to row-major from the transposed cache)
                v_rm = torch.empty((BH, S, hd), **bf)
                ops.transpose16(vtc, smax, hd, S, v_rm, hd, BH, hd * smax, S * hd)
            dsum = torch.empty((BH, S), **f32)
            dqkv = torch.empty((rows, 3 * H), **bf)
            tw = self.twins.get(pre + "wo")
            if self.attn_glue_fused and tw is not None and fused_rows and hd == 128:
                # round 6: d(att) leaves the o_proj dX product as bf16, token-major, and is read like that by the attention backward, whose
                # epilogues write d(q | k | v) with the RoPE backward applied: no fp32 d(att), split16, split_heads16, fp32 dq / dk / dv
                # or rope_merge_bwd passes
                datt16 = torch.empty((rows, H), **bf)
                ops.gemm16_fragw(dh16, None, tw[1], None, H, H, ops.EPI_OUT16, out_hi=datt16)
                self._dw(dh16, st["att"], g[pre + "wo"], pre + "wo")
                ops.attn_backward_fused(q, kc, v_rm, datt16, st["att"], st["lse"], dsum, B, S, nh, hd, eng.cos, eng.sin, 0, dqkv)
            else:
                self._dx(dh16, L.wo, dtmp)                                  # d(att)
                self._dw(dh16, st["att"], g[pre + "wo"], pre + "wo")
                datt16, _ = ops.split16(dtmp, _BF, want_lo=False, kmult=64)
                dO = torch.empty((BH, S, hd), **bf)
                ops.split_heads16(datt16[:, :H].contiguous() if datt16.shape[1] != H else datt16, B, S, nh, hd, dO)
                dq = torch.empty((BH, S, hd), **f32)
                dk = torch.empty((BH, S, hd), **f32)
                dv = torch.empty((BH, S, hd), **f32)
                ops.attn_backward(q, kc, v_rm, dO, st["att"], st["lse"], dsum, B, S, nh, hd, dq, dk, dv)
                ops.rope_merge_bwd(dq, dk, dv, eng.cos, eng.sin, B, S, nh, hd, 0, dqkv)
            self._dx(dqkv, L.wqkv, dtmp)
            self._dw(dqkv, st["x1"], g[pre + "wqkv"], pre + "wqkv")
            ops.rmsnorm_bwd(st["h_in"], L.ln1, dtmp, d.rms_norm_eps, dh, True, g[pre + "ln1"], dx16=dh16)
            saved[i] = None
            if overlap_allreduce_world > 1:
                self._start_layer_allreduce(i)
        # ---- bottom: projector and the trainable embedding rows ----
        if seg_rows:
            ridx = torch.cat(seg_rows)
            dya = torch.empty((ridx.numel(), H), **f32)
            ops.gather_rows(dh, ridx, dya)
            ops.colsum_add(dya, g["proj_b"])
            dya16, _ = ops.split16(dya, _BF, want_lo=False, kmult=64)
            self._dw(dya16[:, :H], torch.cat(seg_a16, dim=0).contiguous(), g["proj_w"], "proj_w")
        if self.train_embed_all:
            keep = torch.ones((rows,), dtype=torch.bool, device=dev)
            if seg_rows:
                keep[torch.cat(seg_rows)] = False
        elif self.embed_grad_tokens:
            keep = torch.isin(ids_flat, torch.tensor(sorted(self.embed_grad_tokens), dtype=ids_flat.dtype, device=dev))
        else:
            keep = None
        if keep is not None:
            ridx = keep.nonzero().reshape(-1)
            if ridx.numel():
                tmp = torch.empty((ridx.numel(), H), **f32)
                ops.gather_rows(dh, ridx, tmp)
                ops.scatter_add_rows(tmp, ids_flat[ridx].contiguous(), g["embed"])
        self._norm_collect = False
        self.micro_batches += 1
        if ev is not None:
            ev[2].record()
            self.phase_events = ev                      # of the LAST call: bench.py reads forward / backward ms of a micro-batch
        return loss

    # ------------------------------------------------------------------------------------------
    def _layer_span(self, i: int) -> Tuple[int, int]:
        """[start, end) of decoder layer i inside the flat gradient (its six tensors are contiguous)."""
        o0, _ = self._slices[f"layers.{i}.wqkv"]
        o1, n1 = self._slices[f"layers.{i}.ln2"]
        return o0, o1 + n1

    def _start_layer_allreduce(self, i: int) -> None:
        from .. import dist as D

        o0, o1 = self._layer_span(i)
        if not hasattr(self, "_inflight"):
            self._inflight, self._reduced = [], []
        self._inflight.append(D.all_reduce_sum_async(self.flat_grad[o0:o1], getattr(self, "grad_comm", torch.float32), self._stage(o0, o1)))
        self._reduced.append((o0, o1))

    def _stage(self, o0: int, o1: int):
        """Slice [o0, o1) of the persistent reduced-precision transport buffer (allocated once, on the first exchange that needs
        it: 2 bytes per gradient element, 13.5 GB at 7B); None when the gradients travel as fp32."""
        comm = getattr(self, "grad_comm", torch.float32)
        if comm == torch.float32:
            return None
        st = getattr(self, "_flat_stage", None)
        if st is None or st.dtype != comm or st.numel() != self.flat_grad.numel():
            st = torch.empty(self.flat_grad.numel(), dtype=comm, device=self.flat_grad.device)
            self._flat_stage = st
        return st[o0:o1]

    def _finalize_grads(self) -> None:
        """A matrix whose dW product never ran since zero_grad() (e.g. mm_projector on a text-only batch) still holds the
        previous step's values: zero it before anyone reads the gradients."""
        for name in getattr(self, "_fresh", ()):
            self.grads[name].zero_()
        self._fresh = set()

    def allreduce_grads(self, world: int, bucket_elems: int = 64 * 1024 * 1024) -> None:
        """Sum gradients over the data-parallel ranks (RCCL on GPUs): large flat buckets; the division by the world
        size is folded into the optimizer's grad_scale.  Slices already exchanged during the backward
        (``overlap_allreduce_world``) are skipped; everything in flight is waited for."""
        self._finalize_grads()
        done = sorted(getattr(self, "_reduced", []))
        works = list(getattr(self, "_inflight", []))
        self._inflight, self._reduced = [], []
        timed = getattr(self, "time_exchange", False) and self.flat_grad.is_cuda
        if timed:                                        # everything between these two events is exchange the backward did not hide
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if world > 1:
            from .. import dist as D

            self._norm_spans = []                        # the exchange changes every gradient: sums of squares taken before it are stale
            comm = getattr(self, "grad_comm", torch.float32)
            n = self.flat_grad.numel()
            pos = 0
            for a, b in done + [(n, n)]:                  # the gaps between the spans reduced so far
                for o in range(pos, a, bucket_elems):
                    e = min(o + bucket_elems, a)
                    works.append(D.all_reduce_sum_async(self.flat_grad[o:e], comm, self._stage(o, e)))
                pos = max(pos, b)
        for w in works:
            w.wait()
        if timed:
            ev1.record()
            self._exchange_events = getattr(self, "_exchange_events", []) + [(ev0, ev1)]

    def exposed_exchange_ms(self) -> float:
        """Sum over the steps since the last call of the compute-stream time spent in allreduce_grads() -- the part of the
        gradient exchange the backward did not overlap (0 on one GPU).  Needs ``time_exchange = True``."""
        evs, self._exchange_events = self._exchange_events, []
        if not evs:
            return 0.0
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in evs))

    def _grad_sumsq(self) -> torch.Tensor:
        """Device double: sum of g^2 over the whole flat gradient (the slices whose dW product already added its share,
        ``last_micro_batch``, are not read again).  Call after ``allreduce_grads``."""
        self._finalize_grads()
        spans = sorted(self._norm_spans)
        if spans:
            acc, first = self._norm_acc, False
        else:
            acc, first = torch.empty((1,), dtype=torch.float64, device=self.flat_grad.device), True
        pos, n = 0, self.flat_grad.numel()
        for a, b in spans + [(n, n)]:                       # the gaps between the spans summed so far
            if a > pos:
                ops.sumsq_f32(self.flat_grad[pos:a], out=acc, accumulate=not first)
                first = False
            pos = max(pos, b)
        self._norm_spans = []
        return acc

    def grad_norm(self, world: int = 1) -> float:
        """L2 norm of the (rank-averaged) gradient over every trainable tensor -- what torch.nn.utils.clip_grad_norm_ returns in HF
        Trainer's step (a host read of the device scalar)."""
        return math.sqrt(float(self._grad_sumsq().item())) / world

    @property
    def last_grad_norm(self) -> Optional[float]:
        """Gradient norm of the last ``step(max_grad_norm=...)`` (read from the device on first access), else None."""
        ss = self._last_sumsq
        if ss is None:
            return None
        if isinstance(ss, tuple):
            self._last_sumsq = math.sqrt(float(ss[0].item())) / ss[1]
        return self._last_sumsq

    def step(self, world: int = 1, max_grad_norm: Optional[float] = None) -> None:
        """AdamW over every trainable tensor (bias-corrected, decoupled weight decay), then zero the gradients.
        ``max_grad_norm``: HF Trainer's gradient clipping (``TrainingArguments.max_grad_norm``, default 1.0, which the reference's
        train_llark.sh leaves alone): gradients are scaled by min(1, max_grad_norm / (norm + 1e-6)).  The squared norm stays in
        device memory and the coefficient is formed inside the AdamW kernels (``llark_adamw_clip``): no host read between the
        backward and the update, no extra pass over the gradients; ``last_grad_norm`` reads it back on demand."""
        if self.flat_m is None:
            raise RuntimeError("this HipLlamaTrainer was built with optimizer_state=False (autograd bridge): use a torch optimizer")
        self._finalize_grads()
        clip = max_grad_norm is not None and max_grad_norm > 0
        sumsq = None
        self._last_sumsq = None
        if clip:
            sumsq = self._grad_sumsq()
            self._last_sumsq = (sumsq.clone(), world)
        else:
            self._norm_spans = []                           # partial sums nobody asked for
        self.step_count += 1
        b1, b2 = self.betas
        for name, p in self.params:
            off, n = self._slices[name]
            tw = self.twins.get(name)
            if tw is not None:
                ops.adamw_twins(p, self.flat_grad[off: off + n], self.flat_m[off: off + n], self.flat_v[off: off + n], self.lr, b1, b2,
                                self.eps, self.wd, self.step_count, 1.0 / world, grad_sumsq=sumsq,
                                max_grad_norm=float(max_grad_norm) if clip else 0.0, wfrag=tw[0], rope_rows=tw[2], wtfrag=tw[1])
                continue
            ops.adamw(p.view(-1), self.flat_grad[off: off + n], self.flat_m[off: off + n], self.flat_v[off: off + n],
                      self.lr, b1, b2, self.eps, 0.0 if p.dim() == 1 else self.wd, self.step_count, 1.0 / world,
                      grad_sumsq=sumsq, max_grad_norm=float(max_grad_norm) if clip else 0.0)
        self._weights_changed()
        self.zero_grad()

    # ------------------------------------------------------------------------------------------
    def export_grads_hf(self) -> Dict[str, torch.Tensor]:
        """Gradients under the reference's state-dict names / layouts (for parity tests and checkpoint tooling)."""
        self._finalize_grads()
        d = self.eng.dims
        H, I = d.hidden_size, d.intermediate_size
        out = {}
        for i in range(d.num_hidden_layers):
            p = f"model.layers.{i}"
            gq = self.grads[f"layers.{i}.wqkv"]
            out[f"{p}.self_attn.q_proj.weight"], out[f"{p}.self_attn.k_proj.weight"], out[f"{p}.self_attn.v_proj.weight"] = (
                gq[:H], gq[H: 2 * H], gq[2 * H:])
            out[f"{p}.self_attn.o_proj.weight"] = self.grads[f"layers.{i}.wo"]
            gg = self.grads[f"layers.{i}.wgu"].view(I // 32, 2, 32, H)
            out[f"{p}.mlp.gate_proj.weight"] = gg[:, 0].reshape(I, H)
            out[f"{p}.mlp.up_proj.weight"] = gg[:, 1].reshape(I, H)
            out[f"{p}.mlp.down_proj.weight"] = self.grads[f"layers.{i}.wdown"]
            out[f"{p}.input_layernorm.weight"] = self.grads[f"layers.{i}.ln1"]
            out[f"{p}.post_attention_layernorm.weight"] = self.grads[f"layers.{i}.ln2"]
        out["model.norm.weight"] = self.grads["norm"]
        out["model.embed_tokens.weight"] = self.grads["embed"]
        if "proj_w" in self.grads:
            out["model.mm_projector.weight"] = self.grads["proj_w"]
            out["model.mm_projector.bias"] = self.grads["proj_b"]
        return out
