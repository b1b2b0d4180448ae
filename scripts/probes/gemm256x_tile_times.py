#!/usr/bin/env python
"""Per-tile time stamps of gemm256x (profiling build, -DG256X_PROF): where a persistent workgroup's time goes between K loops.

  LLARK_HIP_LIB=llark_amd/libllark_hip_prof.so python scripts/probes/gemm256x_tile_times.py [cus ...]

For every role of the prior's products (plain c_fc, producer c_proj2, producer c_proj) and every resident-workgroup count asked for
(256 = the shipped launch; 32 = 4 workgroups per XCD), prints per tile: K-loop time, time to ISSUE the epilogue, and the exposed gap
until the next K loop starts (stores acknowledged + chunk barrier), plus how far apart the workgroups of an XCD end their K loops.
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(cus: int):
    import torch

    prof = torch.zeros(256 * 64 * 4, dtype=torch.int64, device="cuda")
    os.environ["LLARK_G256X_PROF_BUF"] = str(prof.data_ptr())
    os.environ["LLARK_G256X_CUS"] = str(cus)
    from llark_amd import ops

    M, W = 65536, 4800
    g = torch.Generator(device="cuda").manual_seed(0)
    hi = torch.randn(M, W, generator=g, device="cuda").half()
    lo = (torch.randn(M, W, generator=g, device="cuda") * 1e-3).half()
    h = torch.randn(M, W, generator=g, device="cuda")
    ohi, olo = torch.zeros(M, W, dtype=torch.float16, device="cuda"), torch.zeros(M, W, dtype=torch.float16, device="cuda")
    part = torch.zeros(M, 2 * ((W + 255) // 256), 2, device="cuda")
    vec = torch.ones(W, device="cuda")
    out = {}
    for name, n, k in (("c_fc plain", 4800, 4800), ("c_proj2 producer", 4800, 4800), ("c_proj producer", 4800, 1216), ("c_proj2 resid", 4800, 4800)):
        wt = (torch.randn(n, k, generator=g, device="cuda") * 0.02).half()
        bias = torch.zeros(n, device="cuda")
        a_hi, a_lo = hi[:, :k].contiguous(), lo[:, :k].contiguous()
        if "plain" in name:
            fn = lambda: ops.gemm16(a_hi, a_lo, wt, bias, n, ops.EPI_QGELU_SPLIT, out_hi=ohi, out_lo=olo, variant=32)
        elif "resid" in name:
            fn = lambda: ops.gemm16(a_hi, a_lo, wt, bias, n, ops.EPI_RESID, c=h, resid=h, variant=32)
        else:
            fn = lambda: ops.gemm16_ln(a_hi, a_lo, wt, bias, n, ops.EPI_RESID, vec, ln_part=part, c=h, resid=h, out_hi=ohi, out_lo=olo)
        fn()
        h.clamp_(-10, 10)
        torch.cuda.synchronize()
        prof.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        p = prof.view(256, 64, 4).cpu().numpy()
        kloop, issue, gap, skew = [], [], [], []
        ntile = 0
        for b in range(cus):
            t = 0
            while t < 63 and p[b, t, 1] != 0:
                kloop.append((p[b, t, 1] - p[b, t, 0]) / 100.0)          # 100 MHz ticks -> us
                issue.append((p[b, t, 2] - p[b, t, 1]) / 100.0)
                gap.append((p[b, t + 1, 0] - p[b, t, 1]) / 100.0)
                t += 1
            ntile = max(ntile, t)
        for x in range(8):                                               # K-loop end times of the workgroups of XCD x, tile by tile
            for t in range(ntile):
                ends = [p[b, t, 1] for b in range(x, cus, 8) if p[b, t, 1] != 0]
                if len(ends) > 1:
                    skew.append((max(ends) - min(ends)) / 100.0)
        q = lambda v, f: round(float(sorted(v)[int(f * (len(v) - 1))]), 2)
        out[name] = {"launch_ms": round(ms, 4), "tiles_per_wg": ntile, "kloop_us_median": q(kloop, .5), "kloop_us_p90": q(kloop, .9),
                     "epilogue_issue_us_median": q(issue, .5), "gap_to_next_kloop_us_median": q(gap, .5), "gap_p10": q(gap, .1), "gap_p90": q(gap, .9),
                     "xcd_kloop_end_skew_us_median": q(skew, .5) if skew else None, "skew_p90": q(skew, .9) if skew else None,
                     "sum_gap_ms_per_wg": round(sum(gap) / cus / 1e3, 4)}
    print(json.dumps({"cus": cus, "roles": out}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        for cus in [int(a) for a in sys.argv[1:]] or [256, 128, 64, 32]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(cus)], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("cus", cus, "FAILED", r.stderr[-2000:])
                continue
            d = json.loads(line[-1])
            print(f"== resident workgroups {cus} ({cus // 8} per XCD)")
            for k, v in d["roles"].items():
                print(f"  {k:18s} " + " ".join(f"{a}={b}" for a, b in v.items()))
