"""Seeded synthetic Jukebox weights and audio (there are no checkpoints offline; README.md:12 of
the reference: "not accompanied with any trained models").

State-dict key names follow upstream openai/jukebox so a real checkpoint can be loaded through the
same door (``encoders.2.*``, ``bottleneck.level_blocks.2.k``, ``prior.*``, ``y_emb.*``).
Init scales: torch Conv1d default for the VQ-VAE; prior per SURVEY Appendix A.3
(``Conv1D.w ~ N(0, 0.02*init_scale)`` rounded to fp16 like upstream ``fp16_params``).  To keep the
synthetic model numerically *interesting* (non-degenerate codes, O(1) LayerNorm inputs, non-trivial
attention) a few scales are raised and documented below; they are test data, not semantics.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from .hparams import JukeboxHParams


def _conv_init(g: torch.Generator, cout: int, cin: int, k: int):
    bound = 1.0 / math.sqrt(cin * k)
    w = (torch.rand(cout, cin, k, generator=g) * 2 - 1) * bound * math.sqrt(3.0)
    b = (torch.rand(cout, generator=g) * 2 - 1) * bound
    return w.float(), b.float()


def make_vqvae_weights(hps: JukeboxHParams, seed: int = 0, codebook_std: float = 1.0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w: Dict[str, torch.Tensor] = {}
    p = "encoders.2"
    for lb, (down_t, stride_t) in enumerate(zip(hps.downs_t, hps.strides_t)):
        cin0 = 1 if lb == 0 else hps.emb_width
        for i in range(down_t):
            b = f"{p}.level_blocks.{lb}.model.{i}"
            w[f"{b}.0.weight"], w[f"{b}.0.bias"] = _conv_init(g, hps.width, cin0 if i == 0 else hps.width, stride_t * 2)
            for r in range(hps.depth):
                rb = f"{b}.1.model.{r}.model"
                w[f"{rb}.1.weight"], w[f"{rb}.1.bias"] = _conv_init(g, hps.width, hps.width, 3)
                w[f"{rb}.3.weight"], w[f"{rb}.3.bias"] = _conv_init(g, hps.width, hps.width, 1)
        b = f"{p}.level_blocks.{lb}.model.{down_t}"
        w[f"{b}.weight"], w[f"{b}.bias"] = _conv_init(g, hps.emb_width, hps.width, 3)
    # codebook placeholder: N(0, codebook_std).  Random conv weights give encoder outputs with a
    # large per-channel DC offset, so an i.i.d. codebook yields degenerate codes; callers replace
    # it with :func:`init_codebook_from_encodings` (the analogue of upstream ``BottleneckBlock.init_k``,
    # which also samples the codebook from encoder outputs).
    w["bottleneck.level_blocks.2.k"] = (torch.randn(hps.l_bins, hps.emb_width, generator=g) * codebook_std).float()
    return w


def make_prior_weights(hps: JukeboxHParams, seed: int = 1, depth: int = None,
                       w_std: float = None, device="cpu") -> Dict[str, torch.Tensor]:
    """Prior weights.  ``w_std`` defaults to 1/sqrt(fan_in)-like scales instead of upstream's
    0.02*init_scale=0.002 so that attention logits / GELU inputs are O(1) with random weights."""
    g = torch.Generator(device=device).manual_seed(seed)
    _randn = torch.randn

    class _T:      # _T.randn(..., device=device) with the seeded generator of that device
        @staticmethod
        def randn(*shape, generator=None):
            return _randn(*shape, generator=generator, device=device)

    W, S, M = hps.prior_width, hps.n_state, hps.mlp_state
    depth = hps.prior_depth if depth is None else depth
    w: Dict[str, torch.Tensor] = {}
    emb_std = 0.5
    w["prior.x_emb.weight"] = (_T.randn(hps.l_bins, W, generator=g) * emb_std).float()
    w["prior.pos_emb.pos_emb"] = (_T.randn(hps.n_ctx, W, generator=g) * 0.1).float()

    def lin(n_in, n_out, std):
        return (_T.randn(n_in, n_out, generator=g) * std).half(), (_T.randn(n_out, generator=g) * 0.02).float()

    for d in range(depth):
        p = f"prior.transformer._attn_mods.{d}"
        w[f"{p}.ln_0.weight"] = (1.0 + 0.1 * _T.randn(W, generator=g)).float()
        w[f"{p}.ln_0.bias"] = (0.05 * _T.randn(W, generator=g)).float()
        w[f"{p}.ln_1.weight"] = (1.0 + 0.1 * _T.randn(W, generator=g)).float()
        w[f"{p}.ln_1.bias"] = (0.05 * _T.randn(W, generator=g)).float()
        s_in = (1.0 / math.sqrt(W)) if w_std is None else w_std
        # q/k columns get a larger scale so that softmax is not uniform
        cw, cb = lin(W, 3 * S, s_in)
        cw = cw.float()
        cw[:, : 2 * S] *= 2.0
        w[f"{p}.attn.c_attn.w"], w[f"{p}.attn.c_attn.b"] = cw.half(), cb
        w[f"{p}.attn.c_proj.w"], w[f"{p}.attn.c_proj.b"] = lin(S, W, (0.5 / math.sqrt(S)) if w_std is None else w_std)
        w[f"{p}.mlp.c_fc.w"], w[f"{p}.mlp.c_fc.b"] = lin(W, M, s_in)
        w[f"{p}.mlp.c_proj.w"], w[f"{p}.mlp.c_proj.b"] = lin(M, W, (0.5 / math.sqrt(M)) if w_std is None else w_std)
    bow_bins, artist_bins = hps.y_bins
    for name, n in (("bow_genre_emb", bow_bins), ("artist_emb", artist_bins), ("total_length_emb", hps.t_bins),
                    ("absolute_pos_emb", hps.t_bins), ("relative_pos_emb", hps.t_bins)):
        w[f"y_emb.{name}.emb.weight"] = (_T.randn(n, W, generator=g) * 0.1).float()
    return w


def make_jukebox_weights(hps: JukeboxHParams, seed: int = 0, depth: int = None, device="cpu") -> Dict[str, torch.Tensor]:
    """``device`` places the (large) prior tensors; values depend on the device's RNG stream."""
    w = make_vqvae_weights(hps, seed)
    w.update(make_prior_weights(hps, seed + 1, depth, device=device))
    return w


def synthetic_clip(clip_idx: int, seconds: float = 25.0, sr: int = 44100) -> np.ndarray:
    """0.5*sin(2*pi*f*t), f~U(110,880), + 0.1*N(0,1); seed 1234+clip_idx.  Returned
    un-normalised; ``load_audio``-style peak normalisation is applied by the caller
    (jukebox/main.py:41-43)."""
    g = torch.Generator().manual_seed(1234 + clip_idx)
    n = int(round(seconds * sr))
    f = 110.0 + (880.0 - 110.0) * torch.rand(1, generator=g).item()
    t = torch.arange(n, dtype=torch.float64) / sr
    x = 0.5 * torch.sin(2 * math.pi * f * t).float() + 0.1 * torch.randn(n, generator=g)
    return x.numpy().astype(np.float32)


def synthetic_clip_rich(clip_idx: int, seconds: float = 25.0, sr: int = 44100) -> np.ndarray:
    """A second synthetic spectrum (round 4, wide parity fixture): six partials between 55 Hz and 4 kHz with slow
    amplitude modulation, a linear chirp 200 -> 6000 Hz, and noise gated on / off in half-second bursts; seed
    4321 + clip_idx.  Un-normalised like :func:`synthetic_clip`."""
    g = torch.Generator().manual_seed(4321 + clip_idx)
    n = int(round(seconds * sr))
    t = torch.arange(n, dtype=torch.float64) / sr
    x = torch.zeros(n, dtype=torch.float64)
    for _ in range(6):
        f = 55.0 * (4000.0 / 55.0) ** torch.rand(1, generator=g).item()
        amp = 0.05 + 0.25 * torch.rand(1, generator=g).item()
        fm = 0.2 + 3.0 * torch.rand(1, generator=g).item()
        ph = 2 * math.pi * torch.rand(1, generator=g).item()
        x += amp * (0.6 + 0.4 * torch.sin(2 * math.pi * fm * t + ph)) * torch.sin(2 * math.pi * f * t + ph)
    f0, f1 = 200.0, 6000.0
    x += 0.15 * torch.sin(2 * math.pi * (f0 * t + 0.5 * (f1 - f0) / max(seconds, 1e-9) * t * t))
    gate = (torch.rand(int(seconds * 2) + 1, generator=g) < 0.4).double().repeat_interleave(sr // 2)[:n]
    x += 0.2 * gate * torch.randn(n, generator=g, dtype=torch.float64)
    return x.float().numpy().astype(np.float32)


def init_codebook_from_encodings(enc: torch.Tensor, l_bins: int, seed: int = 7, noise: float = 0.25) -> torch.Tensor:
    """Data-dependent codebook like upstream ``BottleneckBlock.init_k`` (sample encoder outputs).

    enc: (emb_width, n_tokens) fp32 encoder outputs of one or more calibration clips (computed by
    whichever encoder the caller is exercising: the HIP one in bench.py, the CPU oracle in tests).
    Returns k (l_bins, emb_width) fp32 = sampled columns + ``noise`` * per-channel std * N(0,1).
    """
    g = torch.Generator().manual_seed(seed)
    enc = enc.detach().float().cpu()
    emb, n = enc.shape
    idx = torch.randint(0, n, (l_bins,), generator=g)
    std = enc.std(dim=1, keepdim=True)
    k = enc[:, idx] + noise * std * torch.randn(emb, l_bins, generator=g)
    return k.t().contiguous().float()
