import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/llark_amd") else os.getcwd())
from llark_amd import ops
m, n, k = 2968, 4096, 4096
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(m, k, generator=g, device="cuda")
hi, lo = ops.split16(a, torch.bfloat16, kmult=64)
w = (torch.randn(n, k, generator=g, device="cuda") * 0.05).bfloat16()
wt = ops.pack_weight16(w, False, torch.bfloat16, kmult=64)
wf = ops.pack_weight16_frag(wt, n)
c0 = torch.zeros(m, n, device="cuda"); c1 = torch.zeros(m, n, device="cuda"); c2 = torch.zeros(m, n, device="cuda")
ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_F32, c=c0, variant=0, stream_k=False)
ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_F32, c=c1, stream_k=True)
ops.gemm16_fragw(hi, None, wf, None, n, k, ops.EPI_F32, c=c2, stream_k=True)
d = (c0 - c1).abs()
print("LLARK_SK_PER", os.environ.get("LLARK_SK_PER"), "max diff", float(d.max()), "bad frac", float((d > 1e-3).float().mean()), "run-to-run equal", bool(torch.equal(c1, c2)))
pt = (d[: 23 * 128].reshape(23, 128, 16, 256) > 1e-3).float().mean(dim=(1, 3))
print("bad fraction per tile, rows 0-3:", [[round(float(x), 2) for x in r] for r in pt[:4]])
# where inside a bad tile?
t = (d[:128, :256] > 1e-3)
print("tile0 bad rows (of 128):", int(t.any(dim=1).sum()), "bad cols (of 256):", int(t.any(dim=0).sum()))
print("tile0 bad by 32-col block:", [int(t[:, i*32:(i+1)*32].sum()) for i in range(8)], "by 32-row block:", [int(t[i*32:(i+1)*32].sum()) for i in range(4)])
