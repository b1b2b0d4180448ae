#!/usr/bin/env python
"""Decode-step Linear shapes of Llama-2-7B at B = 1 (GPU): the LDS-DMA weight-streaming kernel (csrc/gemv_dma.hip) against the MFMA
skinny kernel (gemm.hip), same process.  Prints us per call and TB/s of weight bytes; weights are cycled over several copies so
that no call finds its weights in the 256 MiB Infinity Cache.

    python scripts/bench_gemv.py [split|bf16] [M]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llark_amd import ops  # noqa: E402

split = (sys.argv[1] if len(sys.argv) > 1 else "split") == "split"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
H, I = 4096, 11008
shapes = [("qkv", 3 * H, H, ops.EPI_F32), ("o +resid", H, H, ops.EPI_RESID), ("gate_up swiglu", 2 * I, H, ops.EPI_SWIGLU_SPLIT if split else ops.EPI_SWIGLU16),
          ("down +resid", H, I, ops.EPI_RESID), ("lm_head", 32004, H, ops.EPI_F32)]
for name, n, k, epi in shapes:
    copies = max(2, int(600e6 // (n * k * 2)))
    ws = [(torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16() for _ in range(copies)]
    a = torch.randn(M, k, generator=g, device=dev)
    hi = a.bfloat16()
    lo = (a - hi.float()).bfloat16() if split else None
    c = torch.zeros(M, n, device=dev)
    oh = torch.zeros(M, n // 2, dtype=torch.bfloat16, device=dev)
    ol = torch.zeros_like(oh)
    for dma in (True, False):
        ops.GEMV_DMA = dma

        def fn(i):
            w = ws[i % copies]
            if epi in (ops.EPI_SWIGLU16, ops.EPI_SWIGLU_SPLIT):
                ops.gemm16(hi, lo, w, None, n, epi, out_hi=oh, out_lo=ol if split else None)
            elif epi == ops.EPI_RESID:
                ops.gemm16(hi, lo, w, None, n, epi, c=c, resid=c)
            else:
                ops.gemm16(hi, lo, w, None, n, epi, c=c)
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 40
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"M={M} {'split' if split else 'bf16 '} {name:16s} [{n} x {k}] {'lds-dma' if dma else 'mfma   '}: {us:7.1f} us  {n * k * 2 / us / 1e6:5.2f} TB/s", flush=True)
    del ws
