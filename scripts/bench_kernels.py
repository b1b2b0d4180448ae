#!/usr/bin/env python
"""Micro-benchmarks of the non-GEMM kernels at the BASELINE shapes (run on the GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llark_amd import ops

dev = "cuda"
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["attn", "ln", "res"]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


B, T, W, S, heads, blocks = 8, 8192, 4800, 1200, 8, 128
g = torch.Generator(device=dev).manual_seed(0)
if "attn" in which:
    qkv = torch.randn(B * T, 3 * S, generator=g, device=dev)
    hi = torch.zeros(B * T, 1216, dtype=torch.float16, device=dev)
    lo = torch.zeros_like(hi)
    for pat in (1, 2, 3):
        ms = timeit(lambda: ops.prior_attn(qkv, B, T, S, heads, blocks, pat, hi, lo))
        gb = (qkv.numel() * 4 + 2 * B * T * S * 2) / 1e9
        print(f"prior_attn pattern {pat}: {ms*1e3:8.1f} us  ({gb/ms*1e3/1e3:.2f} TB/s of qkv-read-once + out bytes)", flush=True)
if "ln" in which:
    x = torch.randn(B * T, W, generator=g, device=dev)
    gam = torch.ones(W, device=dev)
    bet = torch.zeros(W, device=dev)
    hi = torch.zeros(B * T, W, dtype=torch.float16, device=dev)
    lo = torch.zeros_like(hi)
    ms = timeit(lambda: ops.layernorm_split(x, gam, bet, 1e-5, hi, lo))
    print(f"layernorm_split: {ms*1e3:8.1f} us  ({x.numel()*8/1e9/ms:.2f} TB/s)", flush=True)
if "res" in which:
    for (t, dil) in [(524288, 1), (524288, 27), (262144, 3), (131072, 9)]:
        x = torch.randn(B, 32, t, generator=g, device=dev)
        w1 = ops.pack_conv_weight(torch.randn(32, 32, 3, generator=g, device=dev) * 0.1)
        w2 = ops.pack_conv_weight(torch.randn(32, 32, 1, generator=g, device=dev) * 0.1)
        b1 = torch.zeros(32, device=dev)
        y = torch.empty_like(x)
        ms = timeit(lambda: ops.resblock(x, w1, b1, w2, b1, dil, out=y))
        print(f"resblock T={t} dil={dil}: {ms*1e3:8.1f} us  ({x.numel()*8/1e9/ms:.2f} TB/s algorithmic, {B*t*8192/1e9/ms:.1f} TFLOP/s)", flush=True)
if "decode" in which:
    import types
    from llark_amd.m2t import bench_support as BS
    for prec in ("split", "bf16"):
        args = types.SimpleNamespace(batch=8, llm_precision=prec)
        wl = BS.LLMWorkload(args, torch.device("cuda"))
        eng = wl.engine
        wl.forward(None)                                    # prefill (fills the KV cache up to 371)
        tok = torch.randint(3, 32000, (8, 1), device=dev)
        import time
        wbytes = 32 * (4 * 4096 * 4096 + 3 * 4096 * 11008) * 2 + 32004 * 4096 * 2
        for graph in (False, "list", True):
            eng.decode_graph = graph is True
            eng.decode_replay = graph == "list"
            for _ in range(3):                                  # graph mode: eager, capture, first replay
                eng.forward_tokens(tok, (), pos0=eng.cur_len, last_only=True)
            torch.cuda.synchronize()
            n = 16
            t0 = time.perf_counter()
            for i in range(n):
                eng.forward_tokens(tok, (), pos0=eng.cur_len, last_only=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"decode step B=8 ({prec}, {'hipGraph replay' if graph is True else ('host launch-list replay' if graph else 'eager launches')}): {dt*1e3:.2f} ms/token-step  ({wbytes/dt/1e12:.2f} TB/s of weight bytes)", flush=True)
        del wl, eng
        torch.cuda.empty_cache()
if "vqvae" in which:
    import numpy as np
    from llark_amd.jukebox.hparams import hparams_5b
    from llark_amd.jukebox.synthetic import make_vqvae_weights
    from llark_amd.jukebox.vqvae import VQVAE
    hps = hparams_5b()
    vq = VQVAE(hps, make_vqvae_weights(hps, 0), dev)
    audio = torch.randn(8, hps.sample_length, generator=g, device=dev)
    ms = timeit(lambda: vq.encode_top(audio), iters=3)
    per_clip = ms / 8
    print(f"VQ-VAE level-2 encode B=8: {ms:.2f} ms  -> {per_clip*1e3:.0f} us/clip; algorithmic 1.42 GB/clip -> {1.42/per_clip:.2f} TB/s "
          f"= {1.42/per_clip/8*100:.1f}% of the 8 TB/s HBM roofline; 41 GFLOP/clip -> {41/per_clip:.1f} TFLOP/s", flush=True)
