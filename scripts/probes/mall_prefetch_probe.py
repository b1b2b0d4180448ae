#!/usr/bin/env python
"""Probe: does a decode-step Linear stream its weights faster when something has just READ them (Infinity-Cache residency)?

The decode step leaves HBM idle during attention and kernel ramps (3.35 ms per token against 2.1 ms at the copy rate).  If a weight
matrix that was touched a moment ago streams markedly faster than a cold one, a side-stream prefetcher that touches the NEXT
launch's weights while the current launch runs would turn that idle time into bandwidth.  Measures, for the three streaming
shapes of a Llama-2-7B layer, the GEMV time (a) cold (600 MB of other data read in between), (b) right after a read of the same
matrix, (c) right after itself.  Medians of 15 repetitions, HIP events around the single launch."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llark_amd import ops


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    junk = torch.randn(150_000_000, device=dev)                      # 600 MB
    H, I = 4096, 11008
    x = torch.randn(1, H, generator=g, device=dev)
    x16, x16l = ops.split16(x, torch.bfloat16, want_lo=True)
    norm_w = torch.ones(H, device=dev)
    out = {}
    for name, n, k in (("qkv 100 MB", 3 * H, H), ("gate_up 180 MB", 2 * I, H), ("lm_head 262 MB", 32000, H)):
        w = (torch.randn(n, k, generator=g, device=dev) * 0.02).to(torch.bfloat16)
        c = torch.empty(1, n, device=dev)

        def gemv():
            ops.gemm16(x16, x16l, w, None, n, ops.EPI_F32, c=c)

        def timed(prep):
            ts = []
            for _ in range(15):
                prep()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gemv()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return round(statistics.median(ts), 1), round(min(ts), 1)

        gemv()
        torch.cuda.synchronize()
        cold = timed(lambda: junk.sum())
        touched = timed(lambda: (junk.sum(), w.view(torch.int32).sum()))
        again = timed(lambda: (junk.sum(), gemv()))
        mb = n * k * 2 / 1e6
        out[name] = {"cold_us": cold, "after_touch_us": touched, "after_itself_us": again,
                     "TBps_cold": round(mb / cold[0], 2), "TBps_after_touch": round(mb / touched[0], 2), "TBps_after_itself": round(mb / again[0], 2)}
        print(name, out[name], flush=True)


if __name__ == "__main__":
    main()
