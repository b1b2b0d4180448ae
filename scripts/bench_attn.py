#!/usr/bin/env python
"""Times the causal attention kernels at the shapes the stages run them (GPU): the training forward (with log-sum-exp) and the
flash-style backward of csrc/attn_bwd.hip, and the fp32-class (hi+lo planes) forward of the inference prefill.

    python scripts/bench_attn.py            # prints one line per shape: ms per call, TFLOP/s (causal flops, MFMA flops only)
"""
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llark_amd import ops  # noqa: E402

HD = 128


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    bf = dict(dtype=torch.bfloat16, device="cuda")
    f32 = dict(dtype=torch.float32, device="cuda")
    for (B, nh, S, split) in [(2, 32, 2048, False), (4, 32, 512, False), (8, 32, 371, False), (8, 32, 371, True)]:
        BH, smax = B * nh, ops.round_up(S, 64)
        Sp = ops.round_up(S, 64)
        g = torch.Generator(device="cuda").manual_seed(0)
        q = torch.randn((B, nh, S, HD), generator=g, **f32).to(torch.bfloat16)
        kc = torch.randn((B, nh, smax, HD), generator=g, **f32).to(torch.bfloat16)
        vtc = torch.randn((B, nh, HD, smax), generator=g, **f32).to(torch.bfloat16)
        att = torch.empty((B * S, nh * HD), **bf)
        pairs = S * (S + 1) / 2 * BH                                  # visible (query, key) pairs
        if split:
            lo = [torch.zeros_like(t) for t in (q, kc, vtc)]
            att_lo = torch.empty_like(att)
            ms = timed(lambda: ops.attn_prefill(q, kc, vtc, B, S, nh, HD, 0, att, lo[0], lo[1], lo[2], att_lo))
            print(f"prefill hi+lo  B={B} nh={nh} S={S}: {ms:.3f} ms  {pairs * 4 * HD * 3 / ms / 1e9:.1f} TFLOP/s (3 MFMA passes)")
            continue
        lse = torch.empty((BH, S), **f32)
        ms = timed(lambda: ops.attn_prefill_lse(q, kc, vtc, B, S, nh, HD, att, lse))
        print(f"forward + lse  B={B} nh={nh} S={S}: {ms:.3f} ms  {pairs * 4 * HD / ms / 1e9:.1f} TFLOP/s")
        dO = (torch.randn((BH, S, HD), generator=g, **f32) * 0.1).to(torch.bfloat16)
        qv = q.view(BH, S, HD)
        v_rm = vtc.view(BH, HD, smax)[:, :, :S].transpose(1, 2).contiguous()
        dq, dk, dv = (torch.empty((BH, S, HD), **f32) for _ in range(3))
        dsum = torch.empty((BH, S), **f32)
        ms = timed(lambda: ops.attn_backward(qv, kc, v_rm, dO, att, lse, dsum, B, S, nh, HD, dq, dk, dv))
        print(f"backward       B={B} nh={nh} S={S}: {ms:.3f} ms  {pairs * 14 * HD / ms / 1e9:.1f} TFLOP/s (7 products: S and dP twice)")


if __name__ == "__main__":
    main()
