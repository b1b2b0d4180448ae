#!/usr/bin/env python
"""Plain bf16 GEMM tile variants on the products of the instruction-tuning step (GPU): forward, dX and dW shapes of one Llama-2-7B
layer at M = tokens per micro-step.

    python scripts/bench_gemm_train.py 11,13 [M]        # variants of llark_gemm16_ex; prints ms and TFLOP/s per product
Variants >= 100 are the B-direct kernel on fragment-major weights (100, 101 = its two tiles, 102 = its own choice); -1 = ops.gemm16's
default (what the trainer gets for an operand without a fragment-major twin); 200 = llark_gemm16_t on the operands as the training
step has them (dX: W contraction-major; dW: both operands contraction-major) -- the time of the transposes it replaces is printed too;
210 .. 213 = llark_gemm16_t_ex variants 0 .. 3 (0 = one LDS stage, 1 = 128x256x32 two stages, 2 = 128x256x64 two stages, 3 = 256x256x64);
220 (dW only) = llark_gemm16_ta_fragw: dY as it stands + X^T fragment-major, the time of llark_pack_frag_t16(X) is printed next to it and
counted in the sum; 221 = the same on the 16x16x32 MFMA shape (llark_gemm16_ta_fragw16 over llark_pack_frag_t16x16).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llark_amd import ops  # noqa: E402

VARIANTS = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "11,13").split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = "cuda"


def timeit(fn, iters=8):
    fn()
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
    H, I = 4096, 11008
    # (name, rows, n, k, epilogue): C[rows][n] = A[rows][k] W[n][k]^T
    shapes = [("fwd qkv", M, 3 * H, H, ops.EPI_F32), ("fwd o +resid", M, H, H, ops.EPI_RESID), ("fwd gate_up", M, 2 * I, H, ops.EPI_F32),
              ("fwd down +resid", M, H, I, ops.EPI_RESID), ("dX qkv", M, H, 3 * H, ops.EPI_F32), ("dX gate_up", M, H, 2 * I, ops.EPI_F32),
              ("dX down", M, I, H, ops.EPI_F32), ("dW qkv (+=)", 3 * H, H, M, ops.EPI_RESID), ("dW gate_up (+=)", 2 * I, H, M, ops.EPI_RESID),
              ("dW down (+=)", H, I, M, ops.EPI_RESID)]
    total = {v: 0.0 for v in VARIANTS}
    ref = {}
    for name, rows, n, k, epi in shapes:
        a = (torch.randn(rows, k, generator=g, device=dev) * 0.5).bfloat16()
        wt = (torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16()
        c = torch.zeros(rows, n, device=dev)
        frag = ops.pack_weight16_frag(wt, n) if any(100 <= v < 200 for v in VARIANTS) else None
        is_dw, is_dx = name.startswith("dW"), name.startswith("dX")
        if any(v >= 200 for v in VARIANTS) and (is_dw or is_dx):
            wkn = wt.t().contiguous()                                   # [k][n]: how the forward stores W (dX) / X (dW)
            akm = a.t().contiguous() if is_dw else None                 # [k][rows]: dY as the backward has it (dW)
            ms_t = timeit(lambda: (ops.transposed16(wkn), ops.transposed16(akm) if is_dw else None))
            print(f"M={M} {name:18s} transposes the old path needs: {ms_t:7.3f} ms")
        for v in VARIANTS:
            if v >= 200 and not (is_dw or is_dx):
                continue
            if v in (220, 221) and not is_dw:
                continue
            if v in (220, 221):
                ms_p = timeit(lambda: ops.pack_frag_t16(wkn, n, chunk16=v == 221))
                xt = ops.pack_frag_t16(wkn, n, chunk16=v == 221)
                total[v] += ms_p
                print(f"M={M} {name:18s} llark_pack_frag_t16{'x16' if v == 221 else ''}(X [{k} x {n}]): {ms_p:7.3f} ms")
            def fn():
                kw = dict(c=c, resid=c) if epi == ops.EPI_RESID else dict(c=c)
                if v in (220, 221):
                    ops.gemm16_ta_fragw(akm, xt, rows, n, k, c, accumulate=epi == ops.EPI_RESID, chunk16=v == 221)
                elif v >= 200:
                    tv = -1 if v == 200 else v - 210
                    if is_dw:
                        ops.gemm16_t(akm, wkn, rows, n, k, True, True, c, accumulate=epi == ops.EPI_RESID, variant=tv)
                    else:
                        ops.gemm16_t(a, wkn, rows, n, k, False, True, c, variant=tv)
                elif v >= 100:
                    ops.gemm16_fragw(a, None, frag, None, n, k, epi, variant={100: 0, 101: 1, 102: -1}[v], **kw)
                else:
                    ops.gemm16(a, None, wt, None, n, epi, variant=v, **kw)
            c.zero_()
            fn()
            torch.cuda.synchronize()
            out = c.clone()
            if name not in ref:
                ref[name] = out
            same = bool((out == ref[name]).all())
            ms = timeit(fn)
            total[v] += ms
            print(f"M={M} {name:18s} [{rows} x {n} x {k}] variant {v}: {ms:7.3f} ms {2.0 * rows * n * k / ms / 1e9:7.1f} TFLOP/s  "
                  f"{'= first variant' if same else 'DIFFERS from first variant: max ' + format((out - ref[name]).abs().max().item(), '.3e')}", flush=True)
    for v in VARIANTS:
        print(f"sum over the layer's products, variant {v}: {total[v]:.3f} ms")


if __name__ == "__main__":
    main()
