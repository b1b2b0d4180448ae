#!/usr/bin/env python3
"""RMSNorm backward (llark_rmsnorm_bwd) on the training step's shapes: rows x 4096 fp32, dx accumulated into the residual gradient.
usage: python scripts/bench_rmsnorm_bwd.py [rows ...]      (GPU)"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from llark_amd import ops  # noqa: E402


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [2048, 4096]
    for rows in rows_list:
        width = 4096
        g = torch.Generator(device="cuda").manual_seed(0)
        nsets = max(2, int(1.5e9 // (rows * width * 12)))         # rotate over > 256 MB so that the Infinity Cache does not hold the operands
        xs = [torch.randn((rows, width), device="cuda", generator=g) for _ in range(nsets)]
        dys = [torch.randn((rows, width), device="cuda", generator=g) for _ in range(nsets)]
        dxs = [torch.zeros((rows, width), device="cuda") for _ in range(nsets)]
        w = torch.randn((width,), device="cuda", generator=g)
        dw = torch.zeros_like(w)
        for i in range(5):
            ops.rmsnorm_bwd(xs[i % nsets], w, dys[i % nsets], 1e-5, dxs[i % nsets], True, dw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        e0.record()
        for i in range(n):
            ops.rmsnorm_bwd(xs[i % nsets], w, dys[i % nsets], 1e-5, dxs[i % nsets], True, dw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        gb = rows * width * 4 * 4 / 1e9                        # x, dy, dx read + dx written
        print(f"rmsnorm_bwd rows={rows} width={width}: {us:.1f} us  {gb / us * 1e3:.2f} TB/s (x, dy, dx in, dx out)")


if __name__ == "__main__":
    main()
