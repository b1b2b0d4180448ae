"""Synthetic Llama-2-7B workload for bench.py (random-init weights of the named architecture,
generated on the GPU layer by layer and packed straight into kernel layout)."""
from __future__ import annotations

import time

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
        eng = HipLlamaEngine(dims, device, max_batch=args.batch, max_seq=448, precision=args.llm_precision)
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
        return {"bound": "mfma", "kernel": "gemm_kernel<bf16%s>" % (",split" if self.engine.split else ""),
                "mfma_passes": 2 if self.engine.split else 1, "achieved": round(achieved, 2), "peak": 2500.0,
                "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4), "traffic": None, "launches": launches,
                "avg_launch_ms": round(ms / launches, 4)}


def build(args, device):
    return LLMWorkload(args, device)


def cpu_baseline(args):
    """Oracle Llama forward on the host: B=1, S=371, `cpu_layers` of 32 layers at full width + lm_head,
    extrapolated to 32 layers."""
    from oracle import llama_ref as LR

    layers = max(1, args.cpu_layers)
    spec = LR.LlamaSpec(num_hidden_layers=layers, vocab_size=VOCAB, audio_start_token=START, audio_end_token=END,
                        audio_patch_token=PATCH)
    w = LR.make_weights(spec, seed=0, std=0.02)
    ids = make_prompt_ids(1)
    aud = torch.randn(1, FRAMES, 4800)
    t0 = time.time()
    LR.forward(w, spec, ids, aud, num_layers=0)
    t_head = time.time() - t0
    t0 = time.time()
    LR.forward(w, spec, ids, aud)
    t_all = time.time() - t0
    t_layers = max(t_all - t_head, 1e-6)
    total = t_head + t_layers / layers * 32
    return total, (f"Llama fwd B=1 S=371 fp32 oracle: embed+projector+lm_head {t_head:.2f}s + {layers} of 32 layers "
                   f"{t_layers:.2f}s extrapolated to 32")
