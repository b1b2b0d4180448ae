#!/usr/bin/env python
"""Interleaved A/B timing of the prior's four GEMM shapes (M = 65536, split fp16): variants 12 / 20 (round-1 kernels), 30
(csrc/gemm256.hip, the M-split LDS ring), 31 (csrc/gemm256n.hip, phases over N -- the default) and 41 (csrc/gemm256_lo8n.hip, fp8 low
plane, opt-in).  Variants 12 / 20 / 30 / 31 are also compared bit for bit on every shape.  Rounds are interleaved inside one process and the median / min per variant is reported
(MI355X guide rule 24); operands are random (rule 25).  Run on the GPU box:  python scripts/bench_gemm256.py [variants] [M]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llark_amd import ops

VARIANTS = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "30,31,41".split(","))]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ROUNDS, ITERS = 5, 4
dev = "cuda"


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [("qkv_f32", 3600, 4800, ops.EPI_F32), ("fc_qgelu", 4800, 4800, ops.EPI_QGELU_SPLIT),
              ("proj_resid", 4800, 1216, ops.EPI_RESID), ("proj2_resid", 4800, 4800, ops.EPI_RESID)]
    hi = torch.randn(M, 4800, generator=g, device=dev).half()
    lo = (torch.randn(M, 4800, generator=g, device=dev) * 1e-3).half()
    c = torch.zeros(M, 4800, device=dev)
    cq = torch.zeros(M, 3600, device=dev)
    ohi = torch.zeros(M, 4800, dtype=torch.float16, device=dev)
    olo = torch.zeros_like(ohi)
    res = {}
    for name, n, k, epi in shapes:
        wt = (torch.randn(n, k, generator=g, device=dev) * 0.02).half()
        bias = torch.zeros(n, device=dev)
        a_hi, a_lo = hi[:, :k].contiguous(), lo[:, :k].contiguous()

        a_lo8 = torch.randint(0, 120, (M, a_hi.shape[1]), device=dev, dtype=torch.uint8)      # finite positive E4M3 codes
        olo8 = torch.zeros(M, 4800, dtype=torch.uint8, device=dev)
        sw = ops.lo8_weight_exponent(wt)
        w8p = ops.pack_weight_lo8(wt, sw)

        def fn(v):
            if v == 41:                                   # fp16 hi pass + one MX-fp8 MFMA for the low plane (csrc/gemm256_lo8n.hip)
                w8 = w8p
                if epi == ops.EPI_QGELU_SPLIT:
                    ops.gemm16_lo8(a_hi, a_lo8, wt, sw, bias, n, ops.EPI_QGELU_SPLIT8, out_hi=ohi, out_lo8=olo8, w8=w8)
                elif epi == ops.EPI_RESID:
                    ops.gemm16_lo8(a_hi, a_lo8, wt, sw, bias, n, epi, c=c, resid=c, w8=w8)
                else:
                    ops.gemm16_lo8(a_hi, a_lo8, wt, sw, bias, n, epi, c=cq, w8=w8)
                return
            if epi == ops.EPI_QGELU_SPLIT:
                ops.gemm16(a_hi, a_lo, wt, bias, n, epi, out_hi=ohi, out_lo=olo, variant=v)
            elif epi == ops.EPI_RESID:
                ops.gemm16(a_hi, a_lo, wt, bias, n, epi, c=c, resid=c, variant=v)
            else:
                ops.gemm16(a_hi, a_lo, wt, bias, n, epi, c=cq, variant=v)

        times = {v: [] for v in VARIANTS}
        outs = {}
        for v in VARIANTS:
            if epi == ops.EPI_RESID:
                c.zero_()
            fn(v)
            torch.cuda.synchronize()
            if v != 41:
                outs[v] = (ohi.clone(), olo.clone()) if epi == ops.EPI_QGELU_SPLIT else (c.clone() if epi == ops.EPI_RESID else cq.clone())
        ref_v = min(outs) if outs else None
        for v, o in outs.items():
            same = all(torch.equal(a, b) for a, b in zip(o, outs[ref_v])) if isinstance(o, tuple) else torch.equal(o, outs[ref_v])
            if v != ref_v:
                print(f"split f16 {name:12s} v{v} vs v{ref_v}: {'bit-identical' if same else 'DIFFERENT'}", flush=True)
                res[f"{name}_v{v}_equals_v{ref_v}"] = bool(same)
        del outs
        for _ in range(ROUNDS):
            for v in VARIANTS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(ITERS):
                    fn(v)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / ITERS)
        for v in VARIANTS:
            med, mn = statistics.median(times[v]), min(times[v])
            tf = 2.0 * M * n * k / med / 1e9
            res[f"{name}_v{v}"] = dict(median_ms=round(med, 4), min_ms=round(mn, 4), tflops_algorithmic=round(tf, 1))
            print(f"split f16 {name:12s} M={M} n={n} k={k} v{v:<3d}: median {med:7.3f} ms  min {mn:7.3f} ms  {tf:7.1f} TF algorithmic "
                  f"({2 * tf:7.1f} issued, {tf / 2500:.3f} of 2.5 PF)", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
