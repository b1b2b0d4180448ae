#!/usr/bin/env python
"""Generates tests/golden/llama7b_full32.npz: the fp32 CPU oracle (oracle/llama_ref.py, pinned to the real reference
wrapper by tests/test_oracle_llama.py) run ONCE in the build container on the configuration bench.py times:
random-init Llama-2-7B (32 layers, bf16-valued weights, vocab 32004) + mm_projector, S = 371
(BOS + <audio_start> + 240 patches + <audio_end> + 128 prompt ids), B = 1, then 64 greedy tokens with the KV cache
(BASELINE configs[2]; m2t/models/llamav2.py:224-234,312,339-365).

    python tests/golden/make_llama7b_golden.py               # ~15 min on 8 vCPUs, ~45 GB RAM

Stored: full-vocabulary logits of sampled prefill rows (incl. the last), the final hidden state's probe rows, the 64
greedy tokens with the top-1 / top-2 logit gap of every step (so a GPU-side mismatch can be told from a near-tie),
and the last-position logits of a few decode steps.  Weights are NOT stored (tests/fulldepth.py regenerates them).
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

import fulldepth as FD  # noqa: E402
from oracle import llama_ref as LR  # noqa: E402

ROWS = (0, 1, 2, 120, 241, 242, 243, 300, 369, 370)
NEW_TOKENS = 64
STEP_LOGITS = (0, 1, 31, 63)


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    spec = FD.llama_spec(32)
    t0 = time.time()
    w = FD.llama_weights_cpu(spec)
    for k in list(w):
        w[k] = w[k].float()                       # bf16 VALUES in fp32 storage: .float() in the oracle becomes free
    print(f"weights: {time.time() - t0:.0f}s", flush=True)
    ids, aud = FD.llama_inputs(1)
    assert ids.shape == (1, 371)
    with torch.no_grad():
        t0 = time.time()
        out = LR.forward(w, spec, ids, aud, return_hidden=True)
        print(f"prefill: {time.time() - t0:.0f}s", flush=True)
        logits = out["logits"][0]
        rows = list(ROWS)
        gold = dict(rows=np.array(rows), logits_rows=logits[rows].numpy().astype(np.float32),
                    logits_maxabs=np.float64(logits.abs().max()), hidden_rows=out["hidden"][0, rows].numpy().astype(np.float32),
                    hidden_maxabs=np.float64(out["hidden"].abs().max()))
        past = out["past_key_values"]
        toks, gaps, step_logits = [], [], {}
        cur = out["logits"][:, -1]
        for t in range(NEW_TOKENS):
            top2 = cur[0].topk(2)
            toks.append(int(top2.indices[0]))
            gaps.append(float(top2.values[0] - top2.values[1]))
            if t in STEP_LOGITS:
                step_logits[t] = cur[0].numpy().astype(np.float32).copy()
            if t == NEW_TOKENS - 1:
                break
            o = LR.forward(w, spec, torch.tensor([[toks[-1]]]), None, past_key_values=past)
            past = o["past_key_values"]
            cur = o["logits"][:, -1]
            print(f"token {t + 1}/{NEW_TOKENS}: {toks[-1]} gap {gaps[-1]:.3e}  {time.time() - t0:.0f}s", flush=True)
    gold.update(tokens=np.array(toks, dtype=np.int32), gaps=np.array(gaps, dtype=np.float32),
                step_idx=np.array(sorted(step_logits)), step_logits=np.stack([step_logits[t] for t in sorted(step_logits)]),
                ids_sha=np.array(FD.sha(ids.numpy())), aud_sha=np.array(FD.sha(aud.numpy())))
    np.savez_compressed(FD.LLAMA_NPZ, **gold)
    print("wrote", FD.LLAMA_NPZ, os.path.getsize(FD.LLAMA_NPZ) / 1e6, "MB; min gap", min(gaps))


if __name__ == "__main__":
    main()
