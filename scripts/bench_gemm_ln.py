#!/usr/bin/env python
"""Interleaved timing of the folded-LayerNorm roles of csrc/gemm256x.hip (llark_gemm16_ln) against the plain products of the same
shapes, M = 65536, split fp16 -- the prior's four products.  One process, rounds interleaved, median / min per form (guide rule 24),
random operands (rule 25).  Also the driver of the --pmc passes in scripts/gpu_runs/r04/pmc_gemm_ln.sh (kernel names carry the role:
gemm256x_kernel<f16, EPI, LN>, LN 1 = consumer, 2 = producer).   python scripts/bench_gemm_ln.py [M] [rounds]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llark_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ITERS = 3
dev = "cuda"


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    W = 4800
    hi = torch.randn(M, W, generator=g, device=dev).half()
    lo = (torch.randn(M, W, generator=g, device=dev) * 1e-3).half()
    h = torch.randn(M, W, generator=g, device=dev)
    qkv = torch.zeros(M, 3648, device=dev)
    ohi, olo = torch.zeros(M, W, dtype=torch.float16, device=dev), torch.zeros(M, W, dtype=torch.float16, device=dev)
    part = torch.zeros(M, 2 * ((W + 255) // 256), 2, device=dev)
    stat = torch.zeros(M, 2, device=dev)
    stat[:, 1] = 1.0
    vec = torch.ones(W, device=dev)
    forms = {}
    for name, n, k in (("c_attn", 3648, 4800), ("c_fc", 4800, 4800), ("c_proj", 4800, 1216), ("c_proj2", 4800, 4800)):
        wt = (torch.randn(n, k, generator=g, device=dev) * 0.02).half()
        bias = torch.zeros(n, device=dev)
        a_hi, a_lo = hi[:, :k].contiguous(), lo[:, :k].contiguous()
        if name == "c_attn":
            forms[name + " plain"] = lambda a_hi=a_hi, a_lo=a_lo, wt=wt, bias=bias, n=n: ops.gemm16(a_hi, a_lo, wt, bias, n, ops.EPI_F32, c=qkv)
            forms[name + " consumer"] = lambda a_hi=a_hi, a_lo=a_lo, wt=wt, bias=bias, n=n: ops.gemm16_ln(a_hi, a_lo, wt, bias, n, ops.EPI_F32, vec, ln_stat=stat, c=qkv)
        elif name == "c_fc":
            forms[name + " plain"] = lambda a_hi=a_hi, a_lo=a_lo, wt=wt, bias=bias, n=n: ops.gemm16(a_hi, a_lo, wt, bias, n, ops.EPI_QGELU_SPLIT, out_hi=ohi, out_lo=olo)
            forms[name + " consumer"] = lambda a_hi=a_hi, a_lo=a_lo, wt=wt, bias=bias, n=n: ops.gemm16_ln(a_hi, a_lo, wt, bias, n, ops.EPI_QGELU_SPLIT, vec, ln_stat=stat, out_hi=ohi, out_lo=olo)
        else:
            forms[name + " plain"] = lambda a_hi=a_hi, a_lo=a_lo, wt=wt, bias=bias, n=n: ops.gemm16(a_hi, a_lo, wt, bias, n, ops.EPI_RESID, c=h, resid=h)
            forms[name + " plain on gemm256x"] = lambda a_hi=a_hi, a_lo=a_lo, wt=wt, bias=bias, n=n: ops.gemm16(a_hi, a_lo, wt, bias, n, ops.EPI_RESID, c=h, resid=h, variant=32)
            forms[name + " producer"] = lambda a_hi=a_hi, a_lo=a_lo, wt=wt, bias=bias, n=n: ops.gemm16_ln(a_hi, a_lo, wt, bias, n, ops.EPI_RESID, vec, ln_part=part, c=h, resid=h, out_hi=ohi, out_lo=olo)
    forms["layernorm kernel"] = lambda: ops.layernorm_split(h, vec, vec, 1e-5, ohi, olo)
    forms["ln_stats_finalize"] = lambda: ops.ln_stats_finalize(part, M, part.shape[1], W, 1e-5, stat)
    times = {k: [] for k in forms}
    for k, fn in forms.items():
        fn()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for k, fn in forms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                fn()
                h.clamp_(-10, 10) if "c_proj" in k else None
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / ITERS)
    out = {k: {"median_ms": round(statistics.median(v), 4), "min_ms": round(min(v), 4)} for k, v in times.items()}
    for k, v in out.items():
        print(f"{k:28s} median {v['median_ms']:.4f} ms  min {v['min_ms']:.4f} ms")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
