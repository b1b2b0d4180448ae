#!/usr/bin/env python
"""GEMM tile-variant exploration on the hot-path shapes (run on the GPU box)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llark_amd import ops

VARIANTS = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5".split(","))]
dev = "cuda"


_FRAG = {}


def G(a_hi, a_lo, wt, bias, n, epi, variant=-1, **kw):
    """variant >= 100: the B-direct kernel on fragment-major weights (packed once per weight tensor)."""
    if variant < 100:
        return ops.gemm16(a_hi, a_lo, wt, bias, n, epi, variant=variant, **kw)
    key = (wt.data_ptr(), n)
    if key not in _FRAG:
        _FRAG[key] = ops.pack_weight16_frag(wt, n)
    if variant == 101 and epi in (ops.EPI_SWIGLU16, ops.EPI_SWIGLU_SPLIT):
        variant = 100
    return ops.gemm16_fragw(a_hi, a_lo, _FRAG[key], bias, n, wt.shape[1], epi, variant={100: 0, 101: 1, 102: -1}[variant], **kw)


def check(variant):
    g = torch.Generator().manual_seed(1)
    m, n, k = 333, 450, 200
    a = torch.randn(m, k, generator=g)
    w = (torch.randn(k, n, generator=g) * 0.1).half()
    b = torch.randn(n, generator=g)
    hi, lo = ops.split16(a.to(dev), torch.float16, kmult=64)
    wt = ops.pack_weight16(w.to(dev), True, torch.float16, kmult=64)
    c = torch.full((m, n), float("nan"), device=dev)
    G(hi, lo, wt, b.to(dev), n, ops.EPI_F32, c=c, variant=variant)
    ref = a.double() @ w.double() + b.double()
    err = ((c.cpu().double() - ref).abs() / (a.abs().double() @ w.abs().double() + 1e-30)).max().item()
    # swiglu bf16
    wb = (torch.randn(n - 2, k, generator=g) * 0.1).bfloat16()   # 448 rows = 7*64
    inter = (n - 2) // 2
    gate, up = wb[:inter], wb[inter:]
    packed = torch.stack([gate.view(-1, 32, k), up.view(-1, 32, k)], dim=1).reshape(n - 2, k).contiguous()
    wts = ops.pack_weight16(packed.to(dev), False, torch.bfloat16, kmult=64)
    ab, _ = ops.split16(a.to(dev), torch.bfloat16, want_lo=False, kmult=64)
    osw = torch.zeros((m, inter), dtype=torch.bfloat16, device=dev)
    G(ab, None, wts, None, n - 2, ops.EPI_SWIGLU16, out_hi=osw, variant=variant)
    abf = a.bfloat16().double()
    refs = torch.nn.functional.silu(abf @ gate.double().t()) * (abf @ up.double().t())
    err2 = ((osw.float().cpu().double() - refs).abs() / (refs.abs() + 1.0)).max().item()
    return err, err2


def timeit(fn, iters=5):
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
    res = {}
    for v in ([] if os.environ.get("LLARK_SKIP_CHECK") else VARIANTS):
        e1, e2 = check(v)
        res[f"check_v{v}"] = (e1, e2)
        print(f"variant {v}: split err/(|A||W|) {e1:.2e}  swiglu rel {e2:.2e}", flush=True)
        assert e1 < 6e-7 and e2 < 2e-2, "variant is WRONG"
    g = torch.Generator(device=dev).manual_seed(0)
    M = 65536
    shapes = [("qkv_f32", 3600, 4800, ops.EPI_F32), ("fc_qgelu", 4800, 4800, ops.EPI_QGELU_SPLIT), ("proj_resid", 4800, 1216, ops.EPI_RESID),
              ("proj2_resid", 4800, 4800, ops.EPI_RESID)]
    hi = torch.randn(M, 4800, generator=g, device=dev).half()
    lo = (torch.randn(M, 4800, generator=g, device=dev) * 1e-3).half()
    c = torch.zeros(M, 4800, device=dev)
    ohi = torch.zeros(M, 4800, dtype=torch.float16, device=dev)
    olo = torch.zeros_like(ohi)
    for name, n, k, epi in shapes:
        wt = (torch.randn(n, k, generator=g, device=dev) * 0.02).half()
        bias = torch.zeros(n, device=dev)
        a_hi, a_lo = hi[:, :k].contiguous(), lo[:, :k].contiguous()
        for v in VARIANTS:
            def fn():
                if epi == ops.EPI_QGELU_SPLIT:
                    G(a_hi, a_lo, wt, bias, n, epi, out_hi=ohi, out_lo=olo, variant=v)
                elif epi == ops.EPI_RESID:
                    G(a_hi, a_lo, wt, bias, n, epi, c=c, resid=c, variant=v)
                else:
                    G(a_hi, a_lo, wt, bias, n, epi, c=c[:, :n].contiguous() if n != 4800 else c, variant=v)
            ms = timeit(fn)
            tf = 2.0 * M * n * k / ms / 1e9
            res[f"{name}_v{v}"] = (ms, tf)
            print(f"split f16 {name:12s} n={n} k={k} variant {v}: {ms:8.3f} ms  {tf:7.1f} TF algorithmic ({2*tf:7.1f} issued)", flush=True)
    # Llama shapes (bf16 single pass), M = 8 x 371
    M2 = 2968
    x = torch.randn(M2, 11008, generator=g, device=dev).bfloat16()
    c2 = torch.zeros(M2, 32004, device=dev)
    h2 = torch.zeros(M2, 4096, device=dev)
    o16 = torch.zeros(M2, 11008, dtype=torch.bfloat16, device=dev)
    for name, n, k, epi in [("qkv", 12288, 4096, ops.EPI_F32), ("o", 4096, 4096, ops.EPI_RESID), ("gate_up", 22016, 4096, ops.EPI_SWIGLU16), ("down", 4096, 11008, ops.EPI_RESID),
                            ("lm_head", 32004, 4096, ops.EPI_F32)]:
        wt = (torch.randn(n, k, generator=g, device=dev) * 0.02).bfloat16()
        a = x[:, :k].contiguous()
        for v in VARIANTS:
            def fn():
                if epi == ops.EPI_SWIGLU16:
                    G(a, None, wt, None, n, epi, out_hi=o16, variant=v)
                elif epi == ops.EPI_RESID:
                    G(a, None, wt, None, n, epi, c=h2, resid=h2, variant=v)
                else:
                    G(a, None, wt, None, n, epi, c=c2[:, :n].contiguous() if n != 32004 else c2, variant=v)
            ms = timeit(fn)
            tf = 2.0 * M2 * n * k / ms / 1e9
            res[f"llama_{name}_v{v}"] = (ms, tf)
            print(f"bf16 {name:10s} n={n} k={k} variant {v}: {ms:8.3f} ms  {tf:7.1f} TF", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
