#!/usr/bin/env python
"""Round 4 (last GPU minutes): the Llama prefill with RoPE / head split / K and V^T cache writes inside the q|k|v GEMM's epilogue
(llark_gemm16_fragw_rope_qkv, LLARK_PREFILL_FUSE_ROPE) against the two-launch path, same engine, same weights, interleaved:
Llama-2-7B widths, 32 layers, B = 8 x S = 371 (bench.py's LLM stage), both precisions.  Prints ms per forward (HIP events, median
of 5 x 3 forwards per mode) and whether the logits are bit-equal.  Run on the GPU box:  python scripts/gpu_runs/r04/rope_fuse_ab.py"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch

from llark_amd.m2t import bench_support as BS

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()


class A:
    batch = 8
    llm_precision = "bf16"


os.environ["LLARK_PREFILL_FUSE_ROPE"] = "auto"          # build the permuted q|k|v fragments; the mode is switched per forward below
wl = BS.LLMWorkload(A, torch.device("cuda", 0), layers=a.layers)
eng = wl.engine


def timed(mode, reps=3):
    eng.fuse_prefill_rope = mode
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = wl.forward(None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


for prec in ("bf16", "split"):
    eng.set_precision(prec)
    for mode in ("0", "auto"):
        timed(mode, 1)
    assert eng._prefill_rope_fused(8, 371)
    ts = {"0": [], "auto": []}
    outs = {}
    for _ in range(a.rounds):
        for mode in ("0", "auto"):
            ms, out = timed(mode)
            ts[mode].append(ms)
            outs[mode] = out.clone()
    same = torch.equal(outs["0"], outs["auto"])
    m0, m1 = statistics.median(ts["0"]), statistics.median(ts["auto"])
    print(f"{prec}: two launches {m0:.3f} ms (min {min(ts['0']):.3f}) | fused epilogue {m1:.3f} ms (min {min(ts['auto']):.3f}) | "
          f"delta {m1 - m0:+.3f} ms = {100 * (m1 / m0 - 1):+.2f} % | logits bit-equal: {same}", flush=True)
