"""Synthetic Llama-2-7B workload for bench.py (random-init weights of the named architecture,
generated on the GPU layer by layer and packed straight into kernel layout)."""
from __future__ import annotations

import math
import os

import torch

from .. import ops
from .engine import HipLlamaEngine, LlamaDims

VOCAB = 32004                      # 32000 + [PAD] + <audio_patch>, <audio_start>, <audio_end> (m2t/train.py:110-124)
PATCH, START, END = 32001, 32002, 32003
FRAMES, PROMPT = 240, 128          # 10 fps x 23.8 s ; 128 prompt tokens => S = 1 + 1 + 240 + 1 + 128 = 371


def make_prompt_ids(batch: int, seed: int = 7) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    rows = []
    for _ in range(batch):
        rows.append([1, START] + [PATCH] * FRAMES + [END] + torch.randint(3, 32000, (PROMPT,), generator=g).tolist())
    return torch.tensor(rows, dtype=torch.int64)


class LLMWorkload:
    def __init__(self, args, device, layers=None):
        dims = LlamaDims(vocab_size=VOCAB)
        if layers:
            dims.num_hidden_layers = layers
        self.dims = dims
        self.batch = args.batch
        eng = HipLlamaEngine(dims, device, max_batch=args.batch, max_seq=512, precision=args.llm_precision)
        g = torch.Generator(device=device).manual_seed(0)
        H, I = dims.hidden_size, dims.intermediate_size

        def n(*shape):
            return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)

        ones = torch.ones(H, device=device)
        for i in range(dims.num_hidden_layers):
            eng.set_layer(i, n(H, H), n(H, H), n(H, H), n(H, H), n(I, H), n(I, H), n(H, I), ones, ones)
        eng.set_globals(n(VOCAB, H), ones, n(VOCAB, H), n(H, dims.mm_hidden_size), torch.zeros(H, device=device))
        self.engine = eng
        self.ids = make_prompt_ids(args.batch).to(device)
        self.device = device
        self._rand_emb = None

    def forward(self, emb):
        """emb: (B, 240, 4800) fp32 from the audio encoder (or None -> N(0,1) stand-in when timing the LLM alone)."""
        if emb is None:
            if self._rand_emb is None:
                g = torch.Generator(device=self.device).manual_seed(3)
                self._rand_emb = torch.randn(self.batch, FRAMES, self.dims.mm_hidden_size, generator=g, device=self.device)
            emb = self._rand_emb
        segs = [(b, 1, emb[b]) for b in range(self.batch)]
        return self.engine.forward_tokens(self.ids, segs)

    def generate(self, emb, new_tokens: int, batch=None, time_decode: bool = False):
        """BASELINE configs[2]: prefill (prompt + audio) then `new_tokens` greedy decode steps against the KV cache
        (argmax on device, stopping criterion disabled for timing -- SURVEY 8d).  batch < self.batch runs the first `batch`
        prompts (the e2e bench attaches a B = 1 generate leg to its line); time_decode brackets the decode steps with one HIP
        event pair (``self.decode_events``)."""
        logits = self.forward_last(emb, batch)
        nxt = logits[:, -1].argmax(-1, keepdim=True)
        out = [nxt]
        if time_decode:
            self.decode_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.decode_events[0].record()
        for _ in range(new_tokens - 1):
            logits = self.engine.forward_tokens(nxt, (), pos0=self.engine.cur_len, last_only=True)
            nxt = logits[:, -1].argmax(-1, keepdim=True)
            out.append(nxt)
        if time_decode:
            self.decode_events[1].record()
        return torch.cat(out, dim=1)

    def forward_last(self, emb, batch=None):
        b = self.batch if batch is None else batch
        if emb is None:
            self.forward(None)                                   # creates the N(0,1) stand-in
            emb = self._rand_emb
        segs = [(i, 1, emb[i]) for i in range(b)]
        return self.engine.forward_tokens(self.ids[:b], segs, last_only=True)

    def decode_weight_bytes(self) -> float:
        d = self.dims
        per_layer = (4 * d.hidden_size * d.hidden_size + 3 * d.hidden_size * d.intermediate_size) * 2
        return per_layer * d.num_hidden_layers + d.vocab_size * d.hidden_size * 2

    def flops_per_step(self) -> float:
        d = self.dims
        rows = self.batch * self.ids.shape[1]
        per_layer = 2.0 * rows * (4 * d.hidden_size * d.hidden_size + 3 * d.hidden_size * d.intermediate_size)
        return per_layer * d.num_hidden_layers + 2.0 * rows * d.hidden_size * d.vocab_size

    def roofline(self, timers, args):
        key = "gemm_split_bf16" if self.engine.split else "gemm_bf16"
        if key not in timers:
            return None
        launches, ms, _ = timers[key]
        achieved = self.flops_per_step() * args.steps / (ms * 1e-3) / 1e12
        traffic = None
        try:                                                          # committed PMC record (7B, 8 x 371 only): average bytes per GEMM launch of a forward
            import json
            import os
            rec = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", "r05_pmc_llm.json")))
            # the record is of ONE shape and kernel build: 8 x 371 positions, Llama-2-7B widths, the default kernels (do not
            # attach it to anything else)
            d = self.dims
            if (self.batch == 8 and d.num_hidden_layers == 32 and self.ids.shape[1] == 371 and d.hidden_size == 4096 and d.intermediate_size == 11008
                    and os.environ.get("LLARK_FRAG", "1") == "1"):
                traffic = rec["per_forward"]["split" if self.engine.split else "bf16"]["traffic_bytes_per_launch_avg"]
        except (OSError, ValueError, KeyError):
            pass
        return {"bound": "mfma", "kernel": "gemm_kernel<bf16%s>" % (",split" if self.engine.split else ""),
                "mfma_passes": 2 if self.engine.split else 1, "achieved": round(achieved, 2), "peak": 2500.0,
                "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4), "traffic": traffic,
                "traffic_note": "RECORDED, not measured in this run: memory-side bytes per GEMM launch (average over the 129 launches of a forward, 8 x 371 positions, 7B widths) from the committed --pmc passes in profiles/r05_pmc_llm.json (register-staged round-5 kernels; the DMA-loop kernels move the same operands)" if traffic else None,
                "traffic_source": "profiles/r05_pmc_llm.json" if traffic else None,
                "launches": launches,
                "avg_launch_ms": round(ms / launches, 4),
                "rope_in_qkv_epilogue": bool(self.engine._prefill_rope_fused(self.batch, self.ids.shape[1])),
                "whole_forward_flops_t": round((self.flops_per_step() + self.attention_flops_per_step()) / 1e12, 2),
                "note": ("GEMM launches only (HIP events); with rope_in_qkv_epilogue the q|k|v launches also rotate q / k and append the K / V^T "
                         "caches (no rope_split_kernel, no fp32 qkv) -- that work is inside the event time, its flops are not counted; "
                         "whole-forward fraction = whole_forward_flops_t / llama forward time / peak")}

    def attention_flops_per_step(self) -> float:
        """Dense-counted attention flops of the prefill (4 S^2 H per layer and clip)."""
        d = self.dims
        s = self.ids.shape[1]
        return 4.0 * s * s * d.hidden_size * d.num_hidden_layers * self.batch


class TrainWorkload:
    """BASELINE configs[3]: instruction-tuning step of random-init Llama-2-7B + projector on frozen (precomputed)
    Jukebox features.  Per GPU: ``batch`` clips per optimizer step = ``micro`` clips x (batch/micro) accumulation
    micro-steps (train_llark.sh: per_device_train_batch_size 2), sequence = 371 prompt+audio positions + answer tokens
    padded to ``seq`` (labels -100 on everything but the answer, like the reference's collator)."""

    def __init__(self, args, device, world, layers=None):
        from .train_engine import HipLlamaTrainer

        dims = LlamaDims(vocab_size=VOCAB)
        if layers:
            dims.num_hidden_layers = layers
        self.dims, self.device, self.world = dims, device, world
        self.micro, self.accum, self.seq = args.micro_batch, max(1, args.batch // args.micro_batch), args.train_seq
        # fused accumulation: ``micro`` holds ``loss_groups`` of the recipe's micro-batches in one pass, the loss normalised per group
        self.loss_groups = max(1, int(getattr(args, "loss_groups", 1)))
        eng = HipLlamaEngine(dims, device, max_batch=self.micro, max_seq=ops.round_up(self.seq, 64), precision="bf16")
        g = torch.Generator(device=device).manual_seed(0)
        H, I = dims.hidden_size, dims.intermediate_size

        def n(*shape):
            return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)

        ones = torch.ones(H, device=device)
        for i in range(dims.num_hidden_layers):
            eng.set_layer(i, n(H, H), n(H, H), n(H, H), n(H, H), n(I, H), n(I, H), n(H, I), ones, ones)
        eng.set_globals(n(VOCAB, H), ones, n(VOCAB, H), n(H, dims.mm_hidden_size), torch.zeros(H, device=device))
        self.engine = eng
        self.trainer = HipLlamaTrainer(eng, lr=5e-5, embed_grad_tokens=[START, END],
                                       grad_comm=torch.bfloat16 if getattr(args, "grad_comm", "fp32") == "bf16" else torch.float32,
                                       gradient_checkpointing=bool(getattr(args, "grad_checkpoint", False)))
        self.trainer.time_exchange = True            # HIP events around allreduce_grads(): the exchange the backward did not hide
        gen = torch.Generator().manual_seed(11 + int(__import__("os").environ.get("RANK", "0")))
        self.batches = []
        for k in range(self.accum):
            ids = make_prompt_ids(self.micro, seed=100 + k)
            ans = torch.randint(3, 32000, (self.micro, self.seq - ids.shape[1]), generator=gen)
            full = torch.cat([ids, ans], dim=1)
            labels = full.clone()
            labels[:, : ids.shape[1]] = -100
            emb = torch.randn(self.micro, FRAMES, dims.mm_hidden_size, generator=gen)
            self.batches.append((full.to(device), labels.to(device), emb.to(device)))

    def step(self):
        tr = self.trainer
        loss = None
        for k, (ids, labels, emb) in enumerate(self.batches):
            segs = [(b, 1, emb[b]) for b in range(self.micro)]
            loss = tr.forward_backward(ids, segs, labels, 1.0 / (self.accum * self.loss_groups),
                                       overlap_allreduce_world=self.world if k == len(self.batches) - 1 else 1,
                                       last_micro_batch=k == len(self.batches) - 1, loss_groups=self.loss_groups)
        tr.allreduce_grads(self.world)
        tr.step(self.world, max_grad_norm=1.0)       # HF Trainer's default clip_grad_norm_: one reduction over the 27 GB of gradients
        return loss

    def flops_per_step(self) -> float:
        d = self.dims
        rows = self.micro * self.accum * self.seq
        per_layer = 2.0 * rows * (4 * d.hidden_size * d.hidden_size + 3 * d.hidden_size * d.intermediate_size)
        # fwd + dX + dW for every layer GEMM; lm_head: fwd + dX only (frozen)
        return 3.0 * per_layer * d.num_hidden_layers + 2.0 * 2.0 * rows * d.hidden_size * d.vocab_size

    def model_flops_per_step(self) -> float:
        """MFU numerator: 6 x (decoder-layer parameters) x tokens + 4 x V x H x tokens -- every layer matrix has forward + dX + dW,
        the frozen lm_head forward + dX only, embed_tokens is a gather (no GEMM); no attention term, no recompute.  (Rounds 1-3
        used the conventional 6 x ALL parameters incl. embedding and lm_head: 2.6 % higher at 7B.)"""
        d = self.dims
        layer_params = d.num_hidden_layers * (4 * d.hidden_size * d.hidden_size + 3 * d.hidden_size * d.intermediate_size)
        return (6.0 * layer_params + 4.0 * d.vocab_size * d.hidden_size) * self.micro * self.accum * self.seq

    def roofline(self, timers, args):
        if "gemm_bf16" not in timers:
            return None
        launches, ms, _ = timers["gemm_bf16"]
        achieved = self.flops_per_step() * args.steps / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "gemm_kernel<bf16> (fwd + dX + dW)", "mfma_passes": 1, "achieved": round(achieved, 2),
                "peak": 2500.0, "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4), "traffic": None, "launches": launches,
                "avg_launch_ms": round(ms / launches, 4)}


def build(args, device):
    return LLMWorkload(args, device)


class ClapWorkload:
    """BASELINE configs[4] audio half: `batch` 10 s 48 kHz clips resident in HBM -> fused log-mel -> HTSAT-base (seeded
    synthetic weights) -> (batch, 512) L2-normalised embeddings (scripts/clap/clap_embeddings.py:63-153)."""

    def __init__(self, args, device, precision="fp32"):
        from ..clap import ClapDims, ClapFrontend, HipClapAudioEncoder, random_state_dict
        from ..clap.frontend import CLIP_SAMPLES

        self.batch = args.batch
        self.dims = ClapDims()
        self.frontend = ClapFrontend(device)
        self.encoder = HipClapAudioEncoder(random_state_dict(self.dims, device=device, seed=0), self.dims, device, precision)
        g = torch.Generator(device=device).manual_seed(3 + int(__import__("os").environ.get("RANK", "0")))
        t = torch.arange(CLIP_SAMPLES, device=device, dtype=torch.float32) / 48000.0
        f0 = 110.0 * (1 + torch.rand(args.batch, 1, generator=g, device=device))
        wav = sum(0.3 / h * torch.sin(2 * math.pi * f0 * h * t[None]) for h in range(1, 9))
        self.wav = (wav * torch.exp(-2.0 * (t % 0.5))[None] + 0.01 * torch.randn(args.batch, CLIP_SAMPLES, generator=g, device=device)).contiguous()

    def embed(self):
        return self.encoder.embed(self.frontend.logmel(self.wav, quantize_int16=True))


class MptWorkload:
    """BASELINE configs[4]: the CLAP HTSAT-base encoder turns `batch` 10 s clips into ONE 512-d frame per clip (the
    reference's scripts/clap embeddings are (1, 512)), which MPT-1B (d_model 2048, 24 blocks, 16 heads, ALiBi, vocab
    50432 + 3) takes through mm_projector into the prompt; prefill + `new_tokens` greedy decode steps."""

    def __init__(self, args, device):
        from .mpt_engine import HipMptEngine, MptDims

        self.batch = args.batch
        V = 50432 + 3
        self.dims = MptDims(vocab_size=V, mm_hidden_size=512)
        self.start, self.end, self.patch = V - 2, V - 1, V - 3
        eng = HipMptEngine(self.dims, device, max_batch=args.batch, max_seq=256, precision=args.llm_precision)
        g = torch.Generator(device=device).manual_seed(0)
        D, E = self.dims.d_model, 4 * self.dims.d_model

        def n(*shape):
            return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)

        sd = {"transformer.wte.weight": n(V, D), "transformer.norm_f.weight": torch.ones(D, device=device),
              "transformer.mm_projector.weight": n(D, 512), "transformer.mm_projector.bias": torch.zeros(D, device=device)}
        for i in range(self.dims.n_layers):
            p = f"transformer.blocks.{i}"
            sd.update({f"{p}.norm_1.weight": torch.ones(D, device=device), f"{p}.norm_2.weight": torch.ones(D, device=device),
                       f"{p}.attn.Wqkv.weight": n(3 * D, D), f"{p}.attn.out_proj.weight": n(D, D),
                       f"{p}.ffn.up_proj.weight": n(E, D), f"{p}.ffn.down_proj.weight": n(D, E)})
        eng.load_state_dict(sd)
        self.engine = eng
        gi = torch.Generator().manual_seed(7)
        rows = [[0, self.start, self.patch, self.end] + torch.randint(3, 50000, (PROMPT,), generator=gi).tolist() for _ in range(args.batch)]
        self.ids = torch.tensor(rows, dtype=torch.int64, device=device)                   # S = 4 + 128 = 132
        self.emb = torch.randn(args.batch, 1, 512, generator=g, device=device)
        self.clap = ClapWorkload(args, device) if getattr(args, "with_clap", False) else None

    def generate(self, new_tokens: int):
        eng = self.engine
        if self.clap is not None:
            self.emb = self.clap.embed().unsqueeze(1)                                # (B, 1, 512), never leaves HBM
        logits = eng.forward_tokens(self.ids, [(b, 1, self.emb[b]) for b in range(self.batch)], last_only=True)
        nxt = logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(new_tokens - 1):
            nxt = eng.forward_tokens(nxt, (), pos0=eng.cur_len)[:, -1].argmax(-1, keepdim=True)
        return nxt

    def prefill_flops(self) -> float:
        d = self.dims
        rows = self.batch * self.ids.shape[1]
        return 2.0 * rows * (12 * d.d_model * d.d_model * d.n_layers + d.d_model * d.vocab_size)


class MptTrainWorkload(MptWorkload):
    """Instruction-tuning step of random-init MPT-1B on CLAP-style (1, 512) embeddings (train_mpt_model.sh analogue of
    BASELINE configs[3]): per GPU `batch` clips x `train_seq` tokens in one micro-batch, fwd + bwd + all-reduce + AdamW."""

    def __init__(self, args, device, world):
        from .mpt_train_engine import HipMptTrainer

        args_bf16 = type("A", (), dict(batch=args.batch, llm_precision="bf16"))()
        super().__init__(args_bf16, device)
        self.world = world
        self.trainer = HipMptTrainer(self.engine, lr=5e-5, grad_comm=torch.bfloat16 if getattr(args, "grad_comm", "fp32") == "bf16" else torch.float32)
        g = torch.Generator().manual_seed(11 + int(__import__("os").environ.get("RANK", "0")))
        ans = torch.randint(3, 50000, (args.batch, args.train_seq - self.ids.shape[1]), generator=g).to(device)
        self.full = torch.cat([self.ids, ans], dim=1)
        self.labels = self.full.clone()
        self.labels[:, : self.ids.shape[1]] = -100
        self.engine.smax = max(self.engine.smax, ops.round_up(args.train_seq, 64))
        self.engine.cos = torch.ones((self.engine.smax, 64), dtype=torch.float32, device=device)
        self.engine.sin = torch.zeros((self.engine.smax, 64), dtype=torch.float32, device=device)
        self.engine.k_cache = None

    def step(self):
        segs = [(b, 1, self.emb[b]) for b in range(self.batch)]
        loss = self.trainer.forward_backward(self.full, segs, self.labels)
        self.trainer.allreduce_grads(self.world)
        self.trainer.step(self.world, max_grad_norm=1.0)
        return loss
