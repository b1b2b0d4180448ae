#!/usr/bin/env python3
"""Greedy-decode step time of the Llama-2-7B engine (random weights), hipGraph replay: B = 1 and B = 8 (run on the GPU box)."""
import sys
import time
import types

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from llark_amd import ops  # noqa: E402
from llark_amd.m2t import bench_support as BS  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "split"
wbytes = 32 * (4 * 4096 * 4096 + 3 * 4096 * 11008) * 2 + 32004 * 4096 * 2
wl = BS.LLMWorkload(types.SimpleNamespace(batch=8, llm_precision=prec), torch.device("cuda"))
eng = wl.engine
for B in (1, 8):
    ids = wl.ids[:B].contiguous()
    emb = torch.randn(B, BS.FRAMES, wl.dims.mm_hidden_size, device="cuda")
    eng.forward_tokens(ids, [(b, 1, emb[b]) for b in range(B)])          # prefill: KV cache up to 371
    tok = torch.randint(3, 32000, (B, 1), device="cuda")
    for _ in range(4):
        eng.forward_tokens(tok, (), pos0=eng.cur_len, last_only=True)
    torch.cuda.synchronize()
    n = 48
    for mode in ("eager", "graph"):                         # wall clock without per-op events
        eng.decode_graph = mode == "graph"
        eng.forward_tokens(ids, [(b, 1, emb[b]) for b in range(B)])      # fresh prefill: the cache holds 512 positions
        for _ in range(4):
            eng.forward_tokens(tok, (), pos0=eng.cur_len, last_only=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.forward_tokens(tok, (), pos0=eng.cur_len, last_only=True)
        torch.cuda.synchronize()
        print(f"decode {prec} B={B} {mode}, no per-op events: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/token", flush=True)
    eng.decode_graph = False
    eng.forward_tokens(ids, [(b, 1, emb[b]) for b in range(B)])
    ops.start_kernel_timing()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_tokens(tok, (), pos0=eng.cur_len, last_only=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    timers = ops.stop_kernel_timing()
    sk = sum(v[1] for k, v in timers.items() if k.endswith("_skinny")) / n
    print(f"decode {prec} B={B}: {dt*1e3:.3f} ms/token  ({wbytes/dt/1e12:.2f} TB/s of weight bytes); skinny GEMMs {sk:.3f} ms/token ({wbytes/sk/1e9:.2f} TB/s)", flush=True)
