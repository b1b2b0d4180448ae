#!/usr/bin/env python
"""Tile-variant sweep of the GEMM on the HTSAT-base linear shapes (64 clips, fp32-class one-launch form: K = 3x).
Run on the GPU box: python scripts/bench_clap_gemm.py [variants]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llark_amd import ops as O

VARIANTS = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "-1,0,1,2,11,12,20".split(","))]
B = int(os.environ.get("CLAP_B", "64"))
dev = "cuda"


def timeit(fn, iters=4):
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
    bf = torch.bfloat16
    total = {v: 0.0 for v in VARIANTS}
    best_total = 0.0
    depth = [2, 2, 12, 2]
    for s in range(4):
        rows, C = B * 4096 // 4 ** s, 128 * 2 ** s
        for name, m, n, k, kind in (("qkv", rows, 3 * C, 3 * C, "f32"), ("proj", rows, C, 3 * C, "resid"), ("fc1", rows, 4 * C, 3 * C, "act"),
                                    ("fc2", rows, C, 12 * C, "resid"), ("red", rows // 4, 2 * C, 12 * C, "f32")):
            if name == "red" and s == 3:
                continue
            a = (torch.randn(m, k, generator=g, device=dev) * 0.5).to(bf)
            w = (torch.randn(n, k, generator=g, device=dev) * 0.05).to(bf)
            bias = torch.randn(n, generator=g, device=dev)
            c = torch.zeros(m, n, device=dev)
            o3 = torch.zeros(m, 3 * n, dtype=bf, device=dev) if kind == "act" else None
            line = f"s{s} {name:5s} m={m:6d} n={n:5d} k={k:5d} "
            times = {}
            for v in VARIANTS:
                if v >= 10 and k % 64:
                    continue
                try:
                    if kind == "f32":
                        fn = lambda: O.gemm16(a, None, w, bias, n, O.EPI_F32, c=c, variant=v)
                    elif kind == "resid":
                        fn = lambda: O.gemm16(a, None, w, bias, n, O.EPI_RESID, c=c, resid=c, variant=v)
                    else:
                        fn = lambda: O.gemm16_act(a, None, w, bias, n, o3[:, :n], o3[:, n:2 * n], o3[:, 2 * n:], act=2, variant=v)
                    times[v] = timeit(fn)
                except Exception as e:  # noqa: BLE001
                    times[v] = float("nan")
            fl = 2.0 * m * n * k
            line += " ".join(f"v{v}:{t * 1e3:7.1f}us({fl / t / 1e9:5.0f}TF)" for v, t in times.items())
            print(line, flush=True)
            mult = depth[s] if name != "red" else 1
            for v, t in times.items():
                total[v] += t * mult
            best_total += min(t for v, t in times.items() if v >= 0 and t == t) * mult
    print("sum over one forward (ms): " + " ".join(f"v{v}:{t:.2f}" for v, t in total.items()) + f" best-per-shape:{best_total:.2f}")


if __name__ == "__main__":
    main()
