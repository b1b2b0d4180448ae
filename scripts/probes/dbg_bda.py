import sys, torch
sys.path.insert(0, "/root/repo")
from llark_amd import ops
m = 256
for k in (256, 512, 1024):
    n = k
    w2 = torch.eye(k).bfloat16()
    wt2 = ops.pack_weight16(w2.cuda(), False, torch.bfloat16, kmult=64); wf2 = ops.pack_weight16_frag(wt2, n)
    aint = (torch.arange(m)[:, None] * 1.0 + torch.arange(k)[None, :] / 1024.0 + 1.0)
    h2, l2 = ops.split16(aint.cuda(), torch.bfloat16, kmult=64)
    for name, (h, l) in (("hi", (h2, torch.zeros_like(l2))), ("lo", (torch.zeros_like(h2), h2))):
        c0 = torch.full((m, n), float("nan"), device="cuda"); c1 = torch.full((m, n), float("nan"), device="cuda")
        ops.gemm16_fragw(h, l, wf2, None, n, k, ops.EPI_F32, c=c0, variant=0, stream_k=False)
        ops.gemm16_fragw(h, l, wf2, None, n, k, ops.EPI_F32, c=c1, variant=2)
        torch.cuda.synchronize()
        bad = (c0 != c1)
        rows = sorted(set((bad.any(1).nonzero().flatten() // 32).tolist()))
        kblocks = sorted(set((bad.any(0).nonzero().flatten() // 16).tolist()))
        print(f"K={k} plane {name}: bad {int(bad.sum())}; bad 32-row blocks {rows}; bad k16 blocks {kblocks}")
        if bad.any():
            r, c = bad.nonzero()[0].tolist()
            print("    first bad (row, k)", r, c, "want", float(c0[r, c]), "got", float(c1[r, c]), "| got as (row', k') guess:", (float(c1[r, c]) - 1.0))
