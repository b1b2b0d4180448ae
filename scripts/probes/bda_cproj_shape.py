#!/usr/bin/env python
"""Feasibility probe (round 5): the prior's c_proj shape (M = 8 x 8192, N = 4800, K = 1216, residual epilogue) on the B-direct DMA loop
(gemm_bda, 128x256 tiles, TWO workgroups per CU: one's epilogue overlaps the other's K loop) against the persistent 256x256 tile
(one workgroup per CU: epilogue serial).  bf16 hi + lo planes stand in for the fp16 ones (same MFMA rate).
    python scripts/probes/bda_cproj_shape.py [M]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llark_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = torch.Generator(device="cuda").manual_seed(0)
forms = {}
for name, n, k in (("c_proj K=1216", 4800, 1216), ("c_proj2 K=4800", 4800, 4800)):
    x = torch.randn(M, k, generator=g, device="cuda")
    w = torch.randn(n, k, generator=g, device="cuda") * 0.02
    c = torch.randn(M, n, generator=g, device="cuda")
    for dt in (torch.bfloat16, torch.float16):
        hi, lo = ops.split16(x, dt, kmult=64)
        w16 = ops.pack_weight16(w, False, dt, kmult=64)
        if dt == torch.bfloat16:
            wf = ops.pack_weight16_frag(w16, n)
            def fn(hi=hi, lo=lo, wf=wf, n=n, k=k, c=c):
                ops.gemm16_fragw(hi, lo, wf, None, n, k, ops.EPI_RESID, c=c, resid=c, variant=2, stream_k=False)
            forms[f"{name} bda bf16 hi+lo RESID"] = (fn, 2.0 * M * n * k)
        else:
            wp = w16
            def fn(hi=hi, lo=lo, wp=wp, n=n, c=c):
                ops.gemm16(hi, lo, wp, None, n, ops.EPI_RESID, c=c, resid=c)
            forms[f"{name} gemm16 fp16 hi+lo RESID (persistent tile)"] = (fn, 2.0 * M * n * k)
# the producer role itself (fp16 planes, predicted statistics): persistent tile (llark_gemm16_ln_p) vs the DMA loop (llark_gemm16_lnp_fragw)
for name, n, k in (("c_proj K=1216", 4800, 1216), ("c_proj2 K=4800", 4800, 4800)):
    x = torch.randn(M, k, generator=g, device="cuda")
    w = torch.randn(n, k, generator=g, device="cuda") * 0.02
    c = torch.randn(M, n, generator=g, device="cuda")
    gamma = torch.ones(n, device="cuda")
    hi, lo = ops.split16(x, torch.float16, kmult=64)
    wt = ops.pack_weight16(w, False, torch.float16, kmult=64)
    wf = ops.pack_weight16_frag(wt, n)
    ph, pl = (torch.zeros((M, n), dtype=torch.float16, device="cuda") for _ in range(2))
    part = torch.zeros((M, 75, 2), device="cuda")
    pred = torch.zeros((M, 2), device="cuda")
    pred[:, 1] = 1.0
    def fx(hi=hi, lo=lo, wt=wt, n=n, c=c, ph=ph, pl=pl, part=part, pred=pred, gamma=gamma):
        ops.gemm16_ln(hi, lo, wt, None, n, ops.EPI_RESID, gamma, ln_part=part, c=c, resid=c, out_hi=ph, out_lo=pl, ln_pred=pred)
    def fb(hi=hi, lo=lo, wf=wf, n=n, k=wt.shape[1], c=c, ph=ph, pl=pl, part=part, pred=pred, gamma=gamma):
        ops.gemm16_lnp_fragw(hi, lo, wf, None, n, k, gamma, part, c, c, ph, pl, ln_pred=pred)
    forms[f"{name} PRODUCER persistent tile (gemm256x)"] = (fx, 2.0 * M * n * k)
    forms[f"{name} PRODUCER DMA loop (gemm_bda_lnp)"] = (fb, 2.0 * M * n * k)
times = {k: [] for k in forms}
for k, (fn, _) in forms.items():
    fn()
torch.cuda.synchronize()
for _ in range(7):
    for k, (fn, _) in forms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 5)
for k, v in times.items():
    med = statistics.median(v)
    print(f"{k:56s} median {med:7.4f} ms  min {min(v):7.4f} ms  {forms[k][1] / (med * 1e-3) / 2.5e15:.3f} of peak")
