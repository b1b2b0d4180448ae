#!/usr/bin/env python
"""Debug: compare the staged lo8 kernel (form chosen by LLARK_LO8_FORM) with the in-register form bit by bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from llark_amd import ops
import test_lo8_gpu as TL

for (m, n, k, epi) in [(4096, 4800, 1216, "resid"), (4096, 4800, 1216, "resid2"), (4096, 4800, 1216, "f32"), (2048, 4800, 4800, "resid"), (2048, 4800, 4800, "f32"), (512, 512, 256, "resid")]:
    a, hi, lo8, wt, sw, ref, exact, mag = TL.operands(m, n, k, seed=5)
    hi_d, lo_d, wt_d = hi.cuda(), lo8.cuda(), wt.cuda()
    w8 = ops.pack_weight_lo8(wt_d, sw)
    h0 = torch.randn(m, n, generator=torch.Generator().manual_seed(2)).cuda()
    outs = []
    for form in ("staged", "staged", "regs"):
        if epi == "resid":
            c = h0.clone()
            ops.gemm16_lo8(hi_d, lo_d, wt_d, sw, None, n, ops.EPI_RESID, c=c, resid=c, w8=w8 if form == "staged" else None)
        elif epi == "resid2":                                  # out of place
            c = torch.zeros(m, n, device="cuda")
            ops.gemm16_lo8(hi_d, lo_d, wt_d, sw, None, n, ops.EPI_RESID, c=c, resid=h0, w8=w8 if form == "staged" else None)
        else:
            c = torch.zeros(m, n, device="cuda")
            ops.gemm16_lo8(hi_d, lo_d, wt_d, sw, None, n, ops.EPI_F32, c=c, w8=w8 if form == "staged" else None)
        torch.cuda.synchronize()
        outs.append(c.cpu())
    want = ref + (h0.cpu().double() if epi.startswith("resid") else 0)
    d01 = (outs[0] != outs[1])
    d02 = (outs[0] != outs[2])
    e0 = (outs[0].double() - want).abs().max().item()
    e2 = (outs[2].double() - want).abs().max().item()
    print(f"{m}x{n}x{k} {epi}: staged run-to-run mismatches {int(d01.sum())}; staged vs regs mismatches {int(d02.sum())} of {d02.numel()}; max|staged - spec| {e0:.3e}, max|regs - spec| {e2:.3e}")
    if d02.any():
        idx = d02.nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        print("   rows%256 hist (32-row bins):", torch.bincount((rows % 256) // 32, minlength=8).tolist(), " cols%256 hist:", torch.bincount((cols % 256) // 32, minlength=8).tolist())
        print("   tile (row/256, col/256) count:", len(set(zip((rows // 256).tolist(), (cols // 256).tolist()))), " max |diff|", (outs[0] - outs[2]).abs().max().item())
        bad = (outs[0].double() - want).abs() > 3e-7 * mag + 1e-5
        print("   staged elements outside the spec tolerance:", int(bad.sum()))
