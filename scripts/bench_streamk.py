#!/usr/bin/env python3
"""Llama-2-7B prefill products at M = 8 x 371 rows: one workgroup per tile vs the stream-K decomposition (run on the GPU box)."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from llark_amd import ops  # noqa: E402

dev = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    res = {}
    cold = "--cold" in sys.argv                     # rotate over enough weight copies that none survives in L2 / Infinity Cache
    for M in ([int(a) for a in sys.argv[1:] if a.isdigit()] or [2968, 371]):
        x = torch.randn(M, 11008, generator=g, device=dev)
        h = torch.zeros(M, 4096, device=dev)
        c = torch.zeros(M, 32004, device=dev)
        o16 = [torch.zeros(M, 11008, dtype=torch.bfloat16, device=dev) for _ in range(2)]
        for name, n, k, epi in [("qkv", 12288, 4096, "f32"), ("o", 4096, 4096, "resid"), ("gate_up", 22016, 4096, "swiglu"),
                                ("down", 4096, 11008, "resid"), ("lm_head", 32004, 4096, "f32")]:
            wt = ops.pack_weight16((torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16(), False, torch.bfloat16, kmult=64)
            wf = ops.pack_weight16_frag(wt, n)
            copies = [wf] + ([wf.clone() for _ in range(max(1, int(600e6 // (wf.numel() * 2))))] if cold else [])
            turn = [0]
            hi, lo = ops.split16(x[:, :k].contiguous(), torch.bfloat16, kmult=64)
            for split in (False, True):
                l = lo if split else None
                for label, kw in [("tile128x256", dict(variant=0, stream_k=False)), ("tile128x128", dict(variant=1, stream_k=False)),
                                  ("stream-k", dict(stream_k=True)), ("library", dict())]:
                    if label == "tile128x128" and epi == "swiglu":
                        continue

                    def fn():
                        turn[0] += 1
                        wf = copies[turn[0] % len(copies)]
                        if epi == "swiglu":
                            ops.gemm16_fragw(hi, l, wf, None, n, k, ops.EPI_SWIGLU_SPLIT if split else ops.EPI_SWIGLU16, out_hi=o16[0],
                                             out_lo=o16[1] if split else None, **kw)
                        elif epi == "resid":
                            ops.gemm16_fragw(hi, l, wf, None, n, k, ops.EPI_RESID, c=h, resid=h, **kw)
                        else:
                            ops.gemm16_fragw(hi, l, wf, None, n, k, ops.EPI_F32, c=c[:, :n].contiguous() if n != 32004 else c, **kw)
                    ms = timeit(fn)
                    tf = 2.0 * M * n * k / ms / 1e9
                    res[f"M{M}_{name}_{'split' if split else 'bf16'}_{label}"] = (round(ms, 4), round(tf, 1))
                    print(f"M={M:5d} {name:8s} n={n:6d} k={k:6d} {'split' if split else 'bf16 '} {label:12s}: {ms:8.4f} ms  {tf:7.1f} TF algorithmic", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
