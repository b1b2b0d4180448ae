#!/usr/bin/env python
"""CPU emulation of the prior's split GEMM arithmetic at FULL depth (36 layers, 5b widths, clip 0), to measure what a
cheaper encoding of the LOW plane costs in accuracy before any kernel is written.

Every Conv1D product of the oracle (oracle/jukebox_ref.py::_conv1d_linear, i.e. upstream transformer/ops.py Conv1D =
addmm with fp32 activations x fp16-valued weights, jukebox/main.py:108 fp16=False) is replaced by

    y = b + hi @ W + dq(lo) @ dq(W)          hi = fp16(x),  lo = x - hi

with dq(.) one of
    f16   lo -> fp16, W exact                       (what csrc/gemm*.hip compute today: two f16 MFMA passes)
    fp8   lo -> e4m3 at a fixed 2^SA scale, W -> e4m3 at a per-matrix 2^SW scale   (v_mfma_scale_f32_32x32x64_f8f6f4)
    fp6   lo, W -> MX e2m3 with an E8M0 scale per 32 consecutive k                  (same instruction, 4x the f16 rate)
    fp6x2 lo = lo_a + lo_b and W = W_a + W_b, each an MX e2m3 plane (the second encodes what the first left); the low product is
          lo_a W_a + lo_b W_a + lo_a W_b (lo_b W_b dropped): 128 + 3 x 32 pipe cycles per 32x32x64 instead of 256
    none  lo dropped                                (single f16 pass: the error the low plane exists to remove)

and the probe rows / pooled embedding are compared with tests/golden/jukebox_full36.npz (the exact-fp32 oracle).

    python scripts/sim_lo_quant_error.py fp8 [threads]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fulldepth as FD  # noqa: E402
from oracle import jukebox_ref as R  # noqa: E402

SA = 12          # fixed activation low-plane scale 2^SA (fp8 mode)


def q_e4m3(x: torch.Tensor) -> torch.Tensor:
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def q_mx_e2m3(x: torch.Tensor) -> torch.Tensor:
    """MX block format along the LAST dim: 32-element blocks, E8M0 shared scale = 2^(floor(log2(max)) - 2), e2m3 elements."""
    shp = x.shape
    k = shp[-1]
    pad = (-k) % 32
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    xb = x.reshape(-1, 32)
    amax = xb.abs().amax(dim=1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(1e-38))) - 2.0
    scale = torch.exp2(e)
    y = (xb / scale).clamp(-7.5, 7.5)
    a = y.abs()
    step = torch.where(a < 2.0, torch.full_like(a, 0.125), torch.where(a < 4.0, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
    q = torch.round(y / step) * step
    q = q.clamp(-7.5, 7.5) * scale
    q = torch.where(amax > 0, q, torch.zeros_like(q))
    return q.reshape(*shp[:-1], k + pad)[..., :k]


_wcache = {}


def make_linear(mode: str):
    def lin(x, wm, b, dtype=torch.float32):
        size_out = (*x.size()[:-1], wm.shape[1])
        x2 = x.reshape(-1, x.size(-1)).float()
        wf = wm.float()
        hi = x2.half().float()
        y = torch.addmm(b.float(), hi, wf)
        lo = x2 - hi
        if mode == "none":
            pass
        elif mode == "f16":
            y = y + lo.half().float() @ wf
        elif mode == "fp8":
            key = id(wm)
            if key not in _wcache:
                sw = float(torch.floor(torch.log2(448.0 / wf.abs().max())))
                _wcache[key] = (q_e4m3(wf * 2.0 ** sw) * 2.0 ** -sw)
            y = y + (q_e4m3(lo * 2.0 ** SA) * 2.0 ** -SA) @ _wcache[key]
        elif mode == "fp6":
            key = id(wm)
            if key not in _wcache:
                _wcache[key] = q_mx_e2m3(wf.t().contiguous()).t().contiguous()      # blocks along k = dim 0 of [n_in][n_out]
            y = y + q_mx_e2m3(lo) @ _wcache[key]
        elif mode == "fp6x2":
            key = id(wm)
            if key not in _wcache:
                wa = q_mx_e2m3(wf.t().contiguous()).t().contiguous()
                wb = q_mx_e2m3((wf - wa).t().contiguous()).t().contiguous()
                _wcache[key] = (wa, wb)
            wa, wb = _wcache[key]
            la = q_mx_e2m3(lo)
            lb = q_mx_e2m3(lo - la)
            y = y + (la + lb) @ wa + la @ wb
        else:
            raise ValueError(mode)
        return y.view(*size_out)
    return lin


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp8"
    torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1))
    gold = np.load(FD.JUKEBOX_NPZ)
    hps = FD.jukebox_hps()
    w = FD.jukebox_weights_cpu(hps)
    R._conv1d_linear = make_linear(mode)
    z = torch.from_numpy(gold["codes"].astype(np.int64))[None]
    x_cond, y_cond = R.get_cond(w, hps)
    h = R.prior_embed(w, z, x_cond, y_cond, hps)
    rows = list(gold["probe_rows"])
    layers = list(gold["probe_layers"])
    t0 = time.time()
    with torch.no_grad():
        for dl in range(hps.prior_depth):
            h = R.prior_layer(w, h, dl, hps)
            _wcache.clear()
            if dl + 1 in layers:
                i = layers.index(dl + 1)
                ref = gold["probes"][i]
                got = h[0, rows].numpy()
                err = np.abs(got - ref).max() / gold["maxabs"][i]
                print(f"[{mode}] layer {dl + 1:2d}: probe rows max|err| / max|h| = {err:.3e}   ({time.time() - t0:.0f}s)", flush=True)
    acts = h[0].float()
    frame_len = int(np.floor((hps.n_ctx / (hps.sample_length / hps.sr)) / 10))
    pooled = R.windowed_average(acts, frame_len)[0].numpy()
    e10 = np.abs(pooled - gold["emb_f10"]).max() / np.abs(gold["emb_f10"]).max()
    e0 = np.abs(acts.mean(0).numpy() - gold["emb_f0"]).max() / np.abs(gold["emb_f0"]).max()
    print(f"[{mode}] embedding f=10: max|err| / max|ref| = {e10:.3e} (absolute {np.abs(pooled - gold['emb_f10']).max():.3e}; bar 1e-4 absolute);  f=0: {e0:.3e}", flush=True)


if __name__ == "__main__":
    main()
