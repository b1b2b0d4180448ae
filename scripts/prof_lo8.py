#!/usr/bin/env python
"""Per-phase cycle breakdown of gemm256_lo8 (profiling build: scripts/build_lo8_prof.sh; run with
LLARK_HIP_LIB=llark_amd/libllark_hip_lo8prof.so).  Prints, averaged over waves: cycles per phase spent issuing the phase's
instruction stream / in the counted vmcnt wait / in the barrier, and the epilogue cycles per tile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

buf = torch.zeros(512 * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["LLARK_LO8_PROF_BUF"] = hex(buf.data_ptr())
from llark_amd import ops  # noqa: E402

M = 65536
g = torch.Generator(device="cuda").manual_seed(0)
KERNEL = sys.argv[1] if len(sys.argv) > 1 else "f16x2n"   # "lo8" (gemm256_lo8n.hip), "f16x2" (gemm256.hip, variant 30) or "f16x2n" (gemm256n.hip, variant 31)
F16 = KERNEL in ("f16x2", "f16x2n")
VAR = 31 if KERNEL == "f16x2n" else 30
for name, n, k, epi in [("qkv_f32", 3600, 4800, ops.EPI_F32), ("fc_qgelu", 4800, 4800, ops.EPI_QGELU_SPLIT8), ("proj_resid", 4800, 1216, ops.EPI_RESID)]:
    hi = torch.randn(M, k, generator=g, device="cuda").half()
    lo8 = torch.randint(0, 120, (M, k), device="cuda", dtype=torch.uint8)
    wt = (torch.randn(n, k, generator=g, device="cuda") * 0.02).half()
    c = torch.zeros(M, 4800, device="cuda")
    ohi = torch.zeros(M, 4800, dtype=torch.float16, device="cuda")
    olo = torch.zeros(M, 4800, dtype=torch.uint8, device="cuda")
    sw = ops.lo8_weight_exponent(wt)
    w8 = ops.pack_weight_lo8(wt, sw)

    lo16 = (torch.randn(M, k, generator=g, device="cuda") * 1e-3).half() if F16 else None
    olo16 = torch.zeros(M, 4800, dtype=torch.float16, device="cuda") if F16 else None

    def run():
        if F16:
            if epi == ops.EPI_QGELU_SPLIT8:
                ops.gemm16(hi, lo16, wt, None, n, ops.EPI_QGELU_SPLIT, out_hi=ohi, out_lo=olo16, variant=VAR)
            elif epi == ops.EPI_RESID:
                ops.gemm16(hi, lo16, wt, None, n, epi, c=c, resid=c, variant=VAR)
            else:
                ops.gemm16(hi, lo16, wt, None, n, epi, c=c, variant=VAR)
            return
        if epi == ops.EPI_QGELU_SPLIT8:
            ops.gemm16_lo8(hi, lo8, wt, sw, None, n, epi, out_hi=ohi, out_lo8=olo, w8=w8)
        elif epi == ops.EPI_RESID:
            ops.gemm16_lo8(hi, lo8, wt, sw, None, n, epi, c=c, resid=c, w8=w8)
        else:
            ops.gemm16_lo8(hi, lo8, wt, sw, None, n, epi, c=c, w8=w8)

    run()
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    b = buf.cpu().view(512, 8, 4).double()
    main, epi_b = b[:256], b[256:]
    phases = main[..., 3].sum()
    tiles = epi_b[..., 3].sum()
    print(f"[{KERNEL}] {name:10s} n={n} k={k}: {e0.elapsed_time(e1):.3f} ms; per phase (cycles, mean over waves): issue {main[..., 0].sum() / phases:.0f}  "
          f"vmcnt-wait {main[..., 1].sum() / phases:.0f}  barrier {main[..., 2].sum() / phases:.0f}  | epilogue {epi_b[..., 0].sum() / tiles:.0f} cycles per tile-wave "
          f"({tiles / 8:.0f} tiles, {phases / tiles / 1:.0f} phases per tile-wave)")
    w0 = main[0]
    print("           workgroup 0 per wave (issue / vmcnt / barrier per phase): " + ", ".join(f"{(w0[i, 0] / w0[i, 3]):.0f}/{(w0[i, 1] / w0[i, 3]):.0f}/{(w0[i, 2] / w0[i, 3]):.0f}" for i in range(8)))
