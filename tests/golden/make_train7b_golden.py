#!/usr/bin/env python
"""Generates tests/golden/train7b_grads.npz: every gradient of the instruction-tuning step (m2t/train.py:53-277 ->
WrappedLlamav2ForCausalLM.forward + shifted CE, m2t/models/llamav2.py:259-337) AT 7B WIDTH -- hidden 4096, 32 heads of 128,
intermediate 11008, vocab 32004, S = 1024 (371 prompt + audio positions, 653 answer tokens), 2 decoder layers -- from torch
autograd over the CPU oracle (oracle/llama_ref.py, pinned to the real reference wrapper by tests/test_oracle_llama.py) evaluated
in the reference's bf16 dtype flow (act_dtype=bfloat16: the recipe trains with --bf16 True, scripts/training/train_llark.sh).

    python tests/golden/make_train7b_golden.py          # ~3 min on 8 vCPUs, ~12 GB RAM

Two layers at full width exercise every backward kernel at the shapes bench.py --stages train runs them (the products are
M = 1024 rows x {4096, 11008, 12288, 22016}-wide, attention over S = 1024 with 32 heads); depth adds nothing a layer does not
already do.  A full gradient set is 1.7 GB, so the fixture keeps, for every gradient tensor: its Frobenius norm, its row sums and
column sums (every element contributes to both: a wrong tile, stripe or K range moves them), and a fixed 64 x 64 sample of
entries; 1-D gradients (norm weights, projector bias) and the two trainable embedding rows are kept whole.
Weights are NOT stored: tests/test_train_gpu.py regenerates them from the same seed (train7b_setup below).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import llama_ref as LR  # noqa: E402

NPZ = os.path.join(HERE, "train7b_grads.npz")
SEQ, FRAMES, PROMPT, LAYERS = 1024, 240, 128, 2
VOCAB, PATCH, START, END = 32004, 32001, 32002, 32003
SAMPLE = 64


def train7b_setup():
    """Spec, bf16-valued weights (seed 0, std 0.02 like bench.py's random init), token ids, labels and audio features.
    Shared by this script and tests/test_train_gpu.py."""
    spec = LR.LlamaSpec(num_hidden_layers=LAYERS, vocab_size=VOCAB, audio_start_token=START, audio_end_token=END, audio_patch_token=PATCH)
    w = {k: v.bfloat16().float() for k, v in LR.make_weights(spec, seed=0, std=0.02).items()}
    g = torch.Generator().manual_seed(1234)
    prompt = [1, START] + [PATCH] * FRAMES + [END] + torch.randint(3, 32000, (PROMPT,), generator=g).tolist()
    answer = torch.randint(3, 32000, (SEQ - len(prompt),), generator=g).tolist()
    ids = torch.tensor([prompt + answer], dtype=torch.int64)
    labels = ids.clone()
    labels[:, : len(prompt)] = -100
    aud = torch.randn(1, FRAMES, spec.mm_hidden_size, generator=g)
    return spec, w, ids, labels, aud


def sample_index(shape):
    """The fixed (rows, cols) of the 64 x 64 sample of a 2-D gradient."""
    g = torch.Generator().manual_seed(shape[0] * 31 + shape[1])
    return (torch.randperm(shape[0], generator=g)[:SAMPLE].sort().values, torch.randperm(shape[1], generator=g)[:SAMPLE].sort().values)


def summarize(name, grad, spec):
    """What the fixture keeps of one gradient tensor (fp32 numpy arrays keyed by '<name>|<what>')."""
    grad = grad.detach().float().cpu()
    out = {}
    if name == "model.embed_tokens.weight":            # only the <audio_start>/<audio_end> rows are trainable (llamav2.py:396-415)
        out[name + "|rows"] = grad[[spec.audio_start_token, spec.audio_end_token]].numpy()
        return out
    if grad.dim() == 1:
        out[name + "|full"] = grad.numpy()
        return out
    r, c = sample_index(grad.shape)
    out[name + "|norm"] = np.float64(grad.double().norm())
    out[name + "|rowsum"] = grad.double().sum(1).float().numpy()
    out[name + "|colsum"] = grad.double().sum(0).float().numpy()
    out[name + "|sample"] = grad[r][:, c].numpy()
    return out


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    t0 = time.time()
    spec, w, ids, labels, aud = train7b_setup()
    print(f"weights + inputs: {time.time() - t0:.0f}s, S = {ids.shape[1]}", flush=True)
    wp = {k: v.clone().requires_grad_(k != "lm_head.weight") for k, v in w.items()}
    t0 = time.time()
    out = LR.forward(wp, spec, ids, aud, labels=labels, act_dtype=torch.bfloat16, round_probs=True)
    loss = out["loss"]
    loss.backward()
    print(f"fwd + bwd: {time.time() - t0:.0f}s, loss {loss.item():.6f}", flush=True)
    gold = {"loss": np.float64(loss.item()), "seq": np.int64(SEQ), "layers": np.int64(LAYERS)}
    for k, v in wp.items():
        if v.grad is not None:
            gold.update(summarize(k, v.grad, spec))
    np.savez_compressed(NPZ, **gold)
    print("wrote", NPZ, round(os.path.getsize(NPZ) / 1e6, 2), "MB,", len(gold), "arrays")


if __name__ == "__main__":
    main()
