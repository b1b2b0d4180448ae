#!/usr/bin/env python
"""Probe (round 5): does the persistent 256x256 tile run the same shape at the same speed on bf16 planes as on fp16 planes?  (Same kernel
template, same instruction count: v_mfma_f32_16x16x32_bf16 against ..._f16 -- a difference is the matrix pipe's power per instruction.)
Random operands of the same distribution; also zero operands (the DVFS give-back of MI355X_MICROARCH.md).   python scripts/probes/mfma_dtype_clock.py"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llark_amd import ops

M, N, K = 65536, 4800, 4800
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, generator=g, device="cuda")
w = torch.randn(N, K, generator=g, device="cuda") * 0.02
forms = {}
for dt in (torch.float16, torch.bfloat16):
    for zero in (False, True):
        hi, lo = ops.split16(x * (0.0 if zero else 1.0), dt, kmult=64)
        wt = ops.pack_weight16(w * (0.0 if zero else 1.0), False, dt, kmult=64)
        oh, ol = (torch.zeros((M, N), dtype=dt, device="cuda") for _ in range(2))
        def fn(hi=hi, lo=lo, wt=wt, oh=oh, ol=ol):
            ops.gemm16(hi, lo, wt, None, N, ops.EPI_QGELU_SPLIT, out_hi=oh, out_lo=ol, variant=32)
        forms[f"{str(dt).split('.')[1]:9s} {'zeros ' if zero else 'random'}"] = fn
times = {k: [] for k in forms}
for fn in forms.values():
    fn()
torch.cuda.synchronize()
for _ in range(7):
    for k, fn in forms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 5)
for k, v in times.items():
    med = statistics.median(v)
    print(f"gemm256x c_fc shape, hi + lo planes {k}: median {med:7.4f} ms  min {min(v):7.4f} ms  {2.0 * M * N * K / (med * 1e-3) / 2.5e15:.3f} of peak (algorithmic)")
