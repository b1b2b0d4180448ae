"""Probe (round 6): the streaming Linear with the 64 KiB ring of the chained launches (8 slots per wave) against the default 128 KiB ring,
standalone (no waiting): does halving the bytes in flight cost steady-state bandwidth?  Llama-2-7B decode shapes, one row."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from llark_amd import ops  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    H, I = 4096, 11008
    bf = torch.bfloat16
    h = torch.randn(1, H, generator=g, device="cuda")
    ln = torch.ones(H, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    for name, n, epi in (("qkv", 3 * H, ops.EPI_F32), ("gate_up", 2 * I, ops.EPI_SWIGLU_SPLIT)):
        wts = [(torch.randn(n, H, generator=g, device="cuda") * 0.02).to(bf) for _ in range(4)]     # rotate weights: no cache reuse between calls
        c = torch.zeros(1, n, device="cuda")
        oh, ol = torch.zeros(1, n // 2, dtype=bf, device="cuda"), torch.zeros(1, n // 2, dtype=bf, device="cuda")
        it = [0]

        def plain():
            w = wts[it[0] % 4]; it[0] += 1
            ops.gemm16_rmsnorm_a(h, ln, 1e-5, w, n, epi, True, c=c if epi == ops.EPI_F32 else None, out_hi=oh, out_lo=ol)

        def ring8():
            w = wts[it[0] % 4]; it[0] += 1
            ops.gemv16_dma_chain(w, n, epi, True, x=h, norm_w=ln, eps=1e-5, c=c if epi == ops.EPI_F32 else None, out_hi=oh, out_lo=ol, signal=cnt[0:1])

        a, b = timed(plain), timed(ring8)
        a2, b2 = timed(plain), timed(ring8)
        mb = n * H * 2 / 1e6
        print(f"{name:8s} {mb:6.1f} MB  default ring {min(a, a2):6.1f} us ({mb / min(a, a2) / 1e3 * 1e3:.2f} TB/s)   8-slot ring {min(b, b2):6.1f} us ({mb / min(b, b2) / 1e3 * 1e3:.2f} TB/s)")
    # o_proj: MFMA skinny kernel (default for 33.5 MB) against the streaming kernel with the 8-slot ring
    w4 = [(torch.randn(H, H, generator=g, device="cuda") * 0.02).to(bf) for _ in range(4)]
    a_hi, a_lo = torch.randn(1, H, generator=g, device="cuda").to(bf), torch.zeros(1, H, dtype=bf, device="cuda")
    hh = torch.zeros(1, H, device="cuda")
    it = [0]

    def skinny():
        w = w4[it[0] % 4]; it[0] += 1
        ops.gemm16(a_hi, a_lo, w, None, H, ops.EPI_RESID, c=hh, resid=hh)

    def ring8o():
        w = w4[it[0] % 4]; it[0] += 1
        ops.gemv16_dma_chain(w, H, ops.EPI_RESID, True, a_hi=a_hi, a_lo=a_lo, c=hh, resid=hh, signal=cnt[1:2])

    a, b = timed(skinny), timed(ring8o)
    print(f"o_proj    33.6 MB  MFMA skinny {a:6.1f} us   streaming, 8-slot ring {b:6.1f} us")


if __name__ == "__main__":
    main()
