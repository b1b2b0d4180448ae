#!/usr/bin/env python
"""Interleaved timing of llark_gemm16_fragw variant 0 (gemm_bd_kernel, A through registers) against variant 2 (csrc/gemm_bda.hip, A by
LDS-DMA + fragment read-ahead) on the Llama-2-7B prefill shapes in the headline flow (hi + lo bf16 planes), M = 8 x 371.  One process,
rounds interleaved, median / min (guide rule 24), random operands.   python scripts/bench_gemm_bda.py [M] [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llark_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2968
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ITERS = 5
g = torch.Generator(device="cuda").manual_seed(0)
H, I = 4096, 11008
x = torch.randn(M, I, generator=g, device="cuda")
hi, lo = ops.split16(x, torch.bfloat16, kmult=64)
forms = {}
for name, n, k, epi in (("qkv (F32)", 3 * H, H, ops.EPI_F32), ("gate_up (SWIGLU_SPLIT)", 2 * I, H, ops.EPI_SWIGLU_SPLIT), ("lm_head (F32)", 32000, H, ops.EPI_F32),
                        ("o_proj (RESID, whole tiles)", H, H, ops.EPI_RESID), ("down (RESID, whole tiles)", H, I, ops.EPI_RESID)):
    w = (torch.randn(n, k, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    wf = ops.pack_weight16_frag(w, n)
    a_hi, a_lo = hi[:, :k].contiguous(), lo[:, :k].contiguous()
    c = torch.zeros(M, n, device="cuda") if epi != ops.EPI_SWIGLU_SPLIT else None
    oh = torch.zeros(M, n // 2, dtype=torch.bfloat16, device="cuda") if epi == ops.EPI_SWIGLU_SPLIT else None
    ol = torch.zeros_like(oh) if oh is not None else None
    for v in (0, 2):
        def fn(v=v, a_hi=a_hi, a_lo=a_lo, wf=wf, n=n, k=k, epi=epi, c=c, oh=oh, ol=ol):
            ops.gemm16_fragw(a_hi, a_lo, wf, None, n, k, epi, c=c, resid=c if epi == ops.EPI_RESID else None, out_hi=oh, out_lo=ol, variant=v, stream_k=False)
        forms[f"split {name} variant {v}"] = (fn, 2.0 * M * n * k)
    pepi = ops.EPI_SWIGLU16 if epi == ops.EPI_SWIGLU_SPLIT else epi
    for v in ((0, 1, 2) if epi == ops.EPI_RESID else (0, 2)):     # plain bf16 operands: variant 0 = gemm_bd_kernel (registers), 1 = 128x128 tiles, 2 = the DMA loop
        def fn(v=v, a_hi=a_hi, wf=wf, n=n, k=k, epi=pepi, c=c, oh=oh):
            ops.gemm16_fragw(a_hi, None, wf, None, n, k, epi, c=c, resid=c if epi == ops.EPI_RESID else None, out_hi=oh, variant=v, stream_k=False)
        forms[f"plain {name} variant {v}"] = (fn, 2.0 * M * n * k)
times = {k: [] for k in forms}
for k, (fn, _) in forms.items():
    fn()
torch.cuda.synchronize()
for _ in range(ROUNDS):
    for k, (fn, _) in forms.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / ITERS)
for k, v in times.items():
    med = statistics.median(v)
    print(f"{k:46s} median {med * 1e3:8.1f} us  min {min(v) * 1e3:8.1f} us   {forms[k][1] / (med * 1e-3) / 1e12:7.1f} TFLOP/s algorithmic = {forms[k][1] / (med * 1e-3) / 2.5e15:.3f} of peak")
