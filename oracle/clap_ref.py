"""CPU ORACLE (test infrastructure, NOT product code) for the CLAP HTSAT audio encoder behind the reference's
``scripts/clap/clap_embeddings.py:63-107``:
``laion_clap.CLAP_Module(enable_fusion=False, amodel="HTSAT-base").model.get_audio_embedding(audio_features)``.

``laion_clap`` is a third-party dependency that is absent from /root/reference (unpinned in ``dataflow-requirements.txt``);
its published HTSAT-Swin algorithm is restated here in plain torch from log-mel features to the L2-normalised 512-d
embedding: BatchNorm over the mel bins -> bicubic stretch of the time axis to ``spec_size * freq_ratio`` frames -> fold to a
``spec_size x spec_size`` image -> 4x4 patch embedding + LayerNorm -> Swin stages (window attention with relative-position
bias, cyclic shift on odd blocks, MLP with exact GELU, patch merging) -> LayerNorm -> mean over tokens -> 2-layer projection
(ReLU) -> L2 normalisation (laion's ``get_audio_embedding``).

PINNING: every stage up to the projection output is pinned against an INDEPENDENT public port of the same model,
``transformers.ClapAudioModelWithProjection`` (installed transformers 5.15), on a small configuration with seeded weights
loaded into both (tests/golden/make_clap_golden.py -> tests/golden/clap_tiny.npz).  Not pinned: the log-mel front end
(torchlibrosa inside laion_clap; HF's numpy ClapFeatureExtractor is the nearest stand-in) and a real checkpoint.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F


@dataclass
class ClapSpec:
    """HTSAT-base of the reference (CLAP_MODEL_CFG, scripts/clap/clap_embeddings.py:109-122) by default."""
    embed_dim: int = 128
    depths: List[int] = field(default_factory=lambda: [2, 2, 12, 2])
    heads: List[int] = field(default_factory=lambda: [4, 8, 16, 32])
    window: int = 8
    spec_size: int = 256
    mel_bins: int = 64
    patch: int = 4
    mlp_ratio: float = 4.0
    proj_dim: int = 512
    ln_eps: float = 1e-5
    bn_eps: float = 1e-5

    @property
    def freq_ratio(self) -> int:
        return self.spec_size // self.mel_bins

    @property
    def out_width(self) -> int:
        return self.embed_dim * 2 ** (len(self.depths) - 1)


def relative_position_index(ws: int) -> torch.Tensor:
    """(ws*ws, ws*ws) index into the (2ws-1)^2 bias table (Swin)."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def mel_to_image(x: torch.Tensor, spec: ClapSpec) -> torch.Tensor:
    """(B, 1, T, mel) normalised log-mel -> (B, 1, spec, spec): stretch T to spec*freq_ratio (bicubic, align_corners), then
    lay the freq_ratio time chunks side by side along the frequency axis."""
    B, C, T, Fq = x.shape
    width = spec.spec_size * spec.freq_ratio
    if T > width or Fq > spec.spec_size // spec.freq_ratio * spec.freq_ratio:
        raise ValueError("the wav size should be less than or equal to the swin input size")
    if T < width:
        x = F.interpolate(x, (width, Fq), mode="bicubic", align_corners=True)
    r = spec.freq_ratio
    x = x.reshape(B, C * r, width // r, Fq).permute(0, 1, 3, 2).contiguous()
    return x.reshape(B, C, Fq * r, width // r)


def _window_partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def _window_reverse(w, ws, H, W):
    C = w.shape[-1]
    return w.view(-1, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, H, W, C)


def swin_block(w: Dict[str, torch.Tensor], p: str, x: torch.Tensor, H: int, W: int, heads: int, ws: int, shift: int, eps: float):
    B, L, C = x.shape
    if min(H, W) <= ws:                                   # window covers the whole map: no partition shift
        shift, ws = 0, min(H, W)
    hd = C // heads
    h = F.layer_norm(x, (C,), w[f"{p}.layernorm_before.weight"], w[f"{p}.layernorm_before.bias"], eps).view(B, H, W, C)
    if shift:
        h = torch.roll(h, (-shift, -shift), (1, 2))
    win = _window_partition(h, ws)                        # (B*nW, ws*ws, C)
    a = f"{p}.attention.self"
    q = F.linear(win, w[f"{a}.query.weight"], w[f"{a}.query.bias"]).view(-1, ws * ws, heads, hd).transpose(1, 2)
    k = F.linear(win, w[f"{a}.key.weight"], w[f"{a}.key.bias"]).view(-1, ws * ws, heads, hd).transpose(1, 2)
    v = F.linear(win, w[f"{a}.value.weight"], w[f"{a}.value.bias"]).view(-1, ws * ws, heads, hd).transpose(1, 2)
    att = q @ k.transpose(-1, -2) / math.sqrt(hd)
    bias = w[f"{a}.relative_position_bias_table"][relative_position_index(ws).view(-1)].view(ws * ws, ws * ws, heads).permute(2, 0, 1)
    att = att + bias.unsqueeze(0)
    if shift:
        hr = (torch.arange(H) >= H - ws).long() + (torch.arange(H) >= H - shift).long()
        wr = (torch.arange(W) >= W - ws).long() + (torch.arange(W) >= W - shift).long()
        region = (hr[None, :, None, None] * 3 + wr[None, None, :, None]).to(x.dtype)
        mw = _window_partition(region, ws).view(-1, ws * ws)
        mask = mw.unsqueeze(1) - mw.unsqueeze(2)
        mask = mask.masked_fill(mask != 0, -100.0)
        nW = mask.shape[0]
        att = (att.view(B, nW, heads, ws * ws, ws * ws) + mask.view(1, nW, 1, ws * ws, ws * ws)).view(-1, heads, ws * ws, ws * ws)
    ctx = (torch.softmax(att, dim=-1) @ v).transpose(1, 2).reshape(-1, ws * ws, C)
    ctx = F.linear(ctx, w[f"{p}.attention.output.dense.weight"], w[f"{p}.attention.output.dense.bias"])
    h = _window_reverse(ctx, ws, H, W)
    if shift:
        h = torch.roll(h, (shift, shift), (1, 2))
    x = x + h.reshape(B, L, C)
    m = F.layer_norm(x, (C,), w[f"{p}.layernorm_after.weight"], w[f"{p}.layernorm_after.bias"], eps)
    m = F.gelu(F.linear(m, w[f"{p}.intermediate.dense.weight"], w[f"{p}.intermediate.dense.bias"]))
    return x + F.linear(m, w[f"{p}.output.dense.weight"], w[f"{p}.output.dense.bias"])


def patch_merge(w, p, x, H, W, eps):
    B, L, C = x.shape
    g = x.view(B, H, W, C)
    g = torch.cat([g[:, r::2, c::2, :] for c in range(2) for r in range(2)], dim=-1).view(B, -1, 4 * C)
    g = F.layer_norm(g, (4 * C,), w[f"{p}.norm.weight"], w[f"{p}.norm.bias"], eps)
    return F.linear(g, w[f"{p}.reduction.weight"])


def forward(w: Dict[str, torch.Tensor], spec: ClapSpec, input_features: torch.Tensor, normalize: bool = True, taps: dict = None):
    """input_features (B, 1, T, mel) log-mel -> (B, proj_dim) audio embedding (L2-normalised like laion's get_audio_embedding;
    ``normalize=False`` gives HF's ``audio_embeds``)."""
    e = "audio_model.audio_encoder"
    bn = lambda n: w[f"{e}.batch_norm.{n}"]
    x = (input_features - bn("running_mean").view(1, 1, 1, -1)) / torch.sqrt(bn("running_var").view(1, 1, 1, -1) + spec.bn_eps)
    x = x * bn("weight").view(1, 1, 1, -1) + bn("bias").view(1, 1, 1, -1)
    img = mel_to_image(x, spec)
    x = F.conv2d(img, w[f"{e}.patch_embed.proj.weight"], w[f"{e}.patch_embed.proj.bias"], stride=spec.patch).flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (spec.embed_dim,), w[f"{e}.patch_embed.norm.weight"], w[f"{e}.patch_embed.norm.bias"], spec.ln_eps)
    if taps is not None:
        taps["patch_embed"] = x
    H = W = spec.spec_size // spec.patch
    for s, depth in enumerate(spec.depths):
        for b in range(depth):
            x = swin_block(w, f"{e}.layers.{s}.blocks.{b}", x, H, W, spec.heads[s], spec.window, 0 if b % 2 == 0 else spec.window // 2, spec.ln_eps)
        if taps is not None:
            taps[f"stage{s}"] = x
        if s < len(spec.depths) - 1:
            x = patch_merge(w, f"{e}.layers.{s}.downsample", x, H, W, spec.ln_eps)
            H, W = H // 2, W // 2
    x = F.layer_norm(x, (spec.out_width,), w[f"{e}.norm.weight"], w[f"{e}.norm.bias"], spec.ln_eps)
    pooled = x.mean(dim=1)
    y = F.linear(pooled, w["audio_projection.linear1.weight"], w["audio_projection.linear1.bias"])
    y = F.linear(F.relu(y), w["audio_projection.linear2.weight"], w["audio_projection.linear2.bias"])
    return F.normalize(y, dim=-1) if normalize else y


def make_weights(spec: ClapSpec, seed: int = 0, std: float = 0.05) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights under the HF ``ClapAudioModelWithProjection`` state-dict names."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * std
    e = "audio_model.audio_encoder"
    w = {f"{e}.batch_norm.weight": 1 + r(spec.mel_bins), f"{e}.batch_norm.bias": r(spec.mel_bins),
         f"{e}.batch_norm.running_mean": r(spec.mel_bins) * 4, f"{e}.batch_norm.running_var": 1 + r(spec.mel_bins).abs() * 4,
         f"{e}.patch_embed.proj.weight": r(spec.embed_dim, 1, spec.patch, spec.patch) * 4, f"{e}.patch_embed.proj.bias": r(spec.embed_dim),
         f"{e}.patch_embed.norm.weight": 1 + r(spec.embed_dim), f"{e}.patch_embed.norm.bias": r(spec.embed_dim)}
    C = spec.embed_dim
    for s, depth in enumerate(spec.depths):
        for b in range(depth):
            p = f"{e}.layers.{s}.blocks.{b}"
            M = int(C * spec.mlp_ratio)
            for nm, shape in (("layernorm_before.weight", None), ("layernorm_after.weight", None)):
                w[f"{p}.{nm}"] = 1 + r(C)
            w[f"{p}.layernorm_before.bias"], w[f"{p}.layernorm_after.bias"] = r(C), r(C)
            w[f"{p}.attention.self.relative_position_bias_table"] = r((2 * spec.window - 1) ** 2, spec.heads[s]) * 4
            for nm in ("query", "key", "value"):
                w[f"{p}.attention.self.{nm}.weight"], w[f"{p}.attention.self.{nm}.bias"] = r(C, C), r(C)
            w[f"{p}.attention.output.dense.weight"], w[f"{p}.attention.output.dense.bias"] = r(C, C), r(C)
            w[f"{p}.intermediate.dense.weight"], w[f"{p}.intermediate.dense.bias"] = r(M, C), r(M)
            w[f"{p}.output.dense.weight"], w[f"{p}.output.dense.bias"] = r(C, M), r(C)
        if s < len(spec.depths) - 1:
            p = f"{e}.layers.{s}.downsample"
            w[f"{p}.reduction.weight"], w[f"{p}.norm.weight"], w[f"{p}.norm.bias"] = r(2 * C, 4 * C), 1 + r(4 * C), r(4 * C)
            C *= 2
    w[f"{e}.norm.weight"], w[f"{e}.norm.bias"] = 1 + r(C), r(C)
    w["audio_projection.linear1.weight"], w["audio_projection.linear1.bias"] = r(spec.proj_dim, C), r(spec.proj_dim)
    w["audio_projection.linear2.weight"], w["audio_projection.linear2.bias"] = r(spec.proj_dim, spec.proj_dim), r(spec.proj_dim)
    return w


# ---------------------------------------------------------------------------------------------------------------------------
# Waveform -> log-mel front end.  In the reference the host side is scripts/clap/clap_embeddings.py:127-153 (48 kHz read,
# int16 round trip, laion_clap get_audio_features with data_truncating="rand_trunc", data_filling="repeatpad" -> a 480000-sample
# waveform) and the spectrogram is computed INSIDE the model by torchlibrosa (Spectrogram n_fft 1024 / hop 480 / hann / center
# reflect / power 2 -> LogmelFilterBank sr 48000, 64 mels, 50..14000 Hz, slaney scale + slaney norm, 10 log10(max(x, 1e-10))).
# Restated in float64 numpy (rfft instead of the DFT-matrix conv1d).  PINNING: against transformers.ClapFeatureExtractor's
# non-fusion path (same published algorithm; tests/golden/clap_mel.npz).
# ---------------------------------------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402

CLAP_SR, CLAP_NFFT, CLAP_HOP, CLAP_CLIP = 48000, 1024, 480, 480000


def quantize_roundtrip(x: np.ndarray) -> np.ndarray:
    """int16_to_float32(float32_to_int16(x)) of laion_clap (clap_embeddings.py:139)."""
    x = np.clip(np.asarray(x, np.float32), -1.0, 1.0)
    return ((x * 32767.0).astype(np.int16) / 32767.0).astype(np.float32)


def fit_length(x: np.ndarray, max_len: int = CLAP_CLIP, offset: int = 0) -> np.ndarray:
    """rand_trunc (crop at `offset`, the reference draws it at random) / repeatpad (whole repeats, then zeros)."""
    if len(x) > max_len:
        return x[offset:offset + max_len]
    if len(x) < max_len:
        x = np.tile(x, int(max_len / len(x)))
        x = np.pad(x, (0, max_len - len(x)))
    return x


def mel_filterbank_slaney(sr=CLAP_SR, n_fft=CLAP_NFFT, n_mels=64, fmin=50.0, fmax=14000.0) -> np.ndarray:
    """librosa.filters.mel (htk=False, norm='slaney'): (n_mels, n_fft/2+1) float64."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]


def logmel(wave: np.ndarray) -> np.ndarray:
    """(n,) waveform -> (frames, 64) log-mel in dB, frames = n // hop + 1."""
    x = np.asarray(wave, np.float64)
    pad = np.pad(x, CLAP_NFFT // 2, mode="reflect")
    n_frames = 1 + (len(pad) - CLAP_NFFT) // CLAP_HOP
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(CLAP_NFFT) / CLAP_NFFT)                   # periodic hann
    idx = np.arange(CLAP_NFFT)[None, :] + CLAP_HOP * np.arange(n_frames)[:, None]
    power = np.abs(np.fft.rfft(pad[idx] * win, axis=1)) ** 2
    mel = power @ mel_filterbank_slaney().astype(np.float32).astype(np.float64).T         # librosa hands out fp32 filters
    return 10.0 * np.log10(np.maximum(mel, 1e-10))
