#!/usr/bin/env python3
"""Diagnose the stream-K GEMM: A = 1, W = 1 -> every output must equal K; print which tiles / values are off."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from llark_amd import ops  # noqa: E402

m, n, k = 2968, 4096, 4096
a = torch.ones(m, k, device="cuda")
w = torch.ones(n, k, device="cuda").bfloat16()
hi, lo = ops.split16(a, torch.bfloat16, kmult=64)
wt = ops.pack_weight16(w, False, torch.bfloat16, kmult=64)
wf = ops.pack_weight16_frag(wt, n)
for sk in (False, None):
    c = torch.zeros(m, n, device="cuda")
    ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_F32, c=c, variant=0, stream_k=sk)
    torch.cuda.synchronize()
    vals, counts = torch.unique(c, return_counts=True)
    print("stream_k", sk, "unique values:", [(float(v), int(ct)) for v, ct in zip(vals[:12], counts[:12])])
    if sk is None:
        t = c[::128, ::256][:, :16]          # one sample per tile
        print(t[:8].int().tolist())
        t2 = c[:128, :256]
        v2, c2 = torch.unique(t2, return_counts=True)
        print("tile 0 values", [(float(v), int(ct)) for v, ct in zip(v2, c2)])
# random data, single tile column comparisons
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(m, k, generator=g, device="cuda")
hi, lo = ops.split16(a, torch.bfloat16, kmult=64)
w = (torch.randn(n, k, generator=g, device="cuda") * 0.05).bfloat16()
wt = ops.pack_weight16(w, False, torch.bfloat16, kmult=64)
wf = ops.pack_weight16_frag(wt, n)
c0 = torch.zeros(m, n, device="cuda"); c1 = torch.zeros(m, n, device="cuda")
ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_F32, c=c0, variant=0, stream_k=False)
ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_F32, c=c1, variant=0)
d = (c0 - c1).abs()
print("random: max diff", float(d.max()), "per-tile max diff (first 8x16 tiles):")
pt = d[: 23 * 128].reshape(23, 128, 16, 256).amax(dim=(1, 3))
print((pt[:8] > 1e-3).int().tolist())
c2 = torch.zeros(m, n, device="cuda")
ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_F32, c=c2, variant=0)
print("run-to-run equal:", bool(torch.equal(c1, c2)))
r = torch.randn(m, n, generator=g, device="cuda")
c3 = r.clone(); c4 = r.clone()
ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_RESID, c=c3, resid=c3, variant=0, stream_k=False)
ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_RESID, c=c4, resid=c4, variant=0)
print("resid: max diff", float((c3 - c4).abs().max()), " vs f32 result + r:", float((c3 - (c0 + r)).abs().max()), float((c4 - (c1 + r)).abs().max()))
