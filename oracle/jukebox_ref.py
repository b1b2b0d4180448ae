"""CPU ORACLE (test infrastructure, NOT product code) for the Jukebox half of LLark's hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``llark_amd``) never does.

PARITY UNPINNED: the arithmetic restated here lives in a third-party dependency that is absent
from /root/reference -- openai/jukebox @ 08efbbc1d4ed1a3cef96e08a931944c8b4d63bb3
(docker/jukebox-embed.dockerfile:50-51) -- and the reference ships no golden vectors, tests or
checkpoints for it (README.md:12).  This file restates the published algorithm of that commit
(jukebox/vqvae/{vqvae,encdec,resnet,bottleneck}.py, jukebox/prior/{prior,autoregressive,
conditioners}.py, jukebox/transformer/{transformer,ops,factored_attention}.py) and anchors it on the
reference's own call sites:

  jukebox/main.py:54-68    get_z               -> :func:`get_z`
  jukebox/main.py:71-98    get_cond            -> :func:`get_cond`
  jukebox/main.py:101-110  get_final_activations -> :func:`get_final_activations`
  jukebox/main.py:113-130  windowed_average    -> :func:`windowed_average`
  jukebox/main.py:133-173  get_acts_from_file  -> :func:`get_acts_from_audio`

All math is fp32 on CPU (torch ops).  ``Conv1D.w`` of the prior is fp16-VALUED (upstream
``fp16_params=True``) and used as fp32 (``prior.forward(..., fp16=False)``, jukebox/main.py:108).

The bit-exact integer oracle for the VQ-VAE codes is the C restatement in ``jukebox_ref.c`` (a
fixed fused-multiply-add order); this file is the tolerance-level restatement it is validated
against, and the floating-point oracle for the prior.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --- constants of jukebox/main.py:10-22 -------------------------------------------------------
JUKEBOX_SAMPLE_RATE = 44100
T = 8192
JUKEBOX_EXPECTED_SAMPLES_LEN = 1048576
JUKEBOX_SAMPLE_SECONDS = JUKEBOX_EXPECTED_SAMPLES_LEN / JUKEBOX_SAMPLE_RATE
ACTS_SAMPLE_RATE = T / JUKEBOX_SAMPLE_SECONDS


@dataclass
class Spec:
    """Hyper-parameters reached by ``load_model('5b')`` (jukebox/main.py:176-200)."""

    sr: int = 44100
    sample_length: int = 1048576          # main.py:194
    # vqvae (level-2 encoder only: main.py:63 keeps zs[-1])
    downs_t: Tuple[int, ...] = (3, 2, 2)
    strides_t: Tuple[int, ...] = (2, 2, 2)
    emb_width: int = 64
    l_bins: int = 2048
    width: int = 32
    depth: int = 4
    dilation_growth_rate: int = 3
    # prior_5b with prior_depth forced to 36 (main.py:198)
    n_ctx: int = 8192
    prior_width: int = 4800
    prior_depth: int = 36
    heads: int = 8
    blocks: int = 128
    m_attn: float = 0.25
    m_mlp: float = 1.0
    init_scale: float = 0.1
    # labels_v3
    y_bins: Tuple[int, int] = (604, 7898)
    t_bins: int = 128
    max_bow_genre_size: int = 5
    min_duration: float = 60.0
    max_duration: float = 600.0
    # main.py:72 conditioning length
    cond_seconds: int = 62

    @property
    def raw_to_tokens(self) -> int:
        r = 1
        for d, s in zip(self.downs_t, self.strides_t):
            r *= s ** d
        return r

    @property
    def n_state(self) -> int:
        return int(self.m_attn * self.prior_width)

    @property
    def block_ctx(self) -> int:
        return self.n_ctx // self.blocks

    def check(self):
        assert self.sample_length == self.n_ctx * self.raw_to_tokens
        assert self.n_state % self.heads == 0
        assert self.n_ctx % self.blocks == 0


def full_spec() -> Spec:
    return Spec()


def tiny_spec() -> Spec:
    """A small twin with the same structure (3 levels, 3 attention patterns) for fast tests."""
    return Spec(sample_length=65536, n_ctx=512, prior_width=192, prior_depth=3, heads=2, blocks=8)


# ---------------------------------------------------------------------------------------------
# VQ-VAE level-2 encoder (jukebox/vqvae/encdec.py, resnet.py)
# ---------------------------------------------------------------------------------------------
def _enc_prefix(level: int = 2) -> str:
    return f"encoders.{level}"


def vqvae_encoder_forward(w: Dict[str, torch.Tensor], x: torch.Tensor, spec: Spec, level: int = 2):
    """``Encoder.forward`` for the top level: x (N,1,T) fp32 -> (N,emb_width,T/raw_to_tokens).

    EncoderConvBlock = down_t x [Conv1d(k=2s,s,p=s//2) ; Resnet1D(depth)] + Conv1d(width->emb,3,1,1);
    ResConv1DBlock   = x + Conv1d(1x1)(ReLU(Conv1d(k3,dil=d,pad=d)(ReLU(x)))), d = growth**i.
    """
    p = _enc_prefix(level)
    for lb, (down_t, stride_t) in enumerate(zip(spec.downs_t, spec.strides_t)):
        filter_t, pad_t = stride_t * 2, stride_t // 2
        for i in range(down_t):
            b = f"{p}.level_blocks.{lb}.model.{i}"
            x = F.conv1d(x, w[f"{b}.0.weight"], w[f"{b}.0.bias"], stride=stride_t, padding=pad_t)
            for r in range(spec.depth):
                d = spec.dilation_growth_rate ** r
                rb = f"{b}.1.model.{r}.model"
                h = F.conv1d(F.relu(x), w[f"{rb}.1.weight"], w[f"{rb}.1.bias"], padding=d, dilation=d)
                h = F.conv1d(F.relu(h), w[f"{rb}.3.weight"], w[f"{rb}.3.bias"])
                x = x + h
        b = f"{p}.level_blocks.{lb}.model.{down_t}"
        x = F.conv1d(x, w[f"{b}.weight"], w[f"{b}.bias"], padding=1)
    return x


def bottleneck_encode(k: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """``BottleneckBlock.encode``: x (N,emb,T) -> codes (N,T) int64 (first minimal index)."""
    N, width, Tt = x.shape
    xf = x.permute(0, 2, 1).contiguous().view(-1, width)
    k_w = k.t()
    distance = torch.sum(xf ** 2, dim=-1, keepdim=True) - 2 * torch.matmul(xf, k_w) + torch.sum(
        k_w ** 2, dim=0, keepdim=True
    )
    _, x_l = torch.min(distance, dim=-1)
    return x_l.view(N, Tt)


def get_z(audio: np.ndarray, w: Dict[str, torch.Tensor], spec: Spec) -> torch.Tensor:
    """jukebox/main.py:54-68 (the reference runs all three level encoders and keeps zs[-1]; the
    level-2 result does not depend on levels 0/1, so only it is computed)."""
    assert len(audio) >= spec.sample_length, (
        f"expected samples with shape {spec.sample_length}; got shape {audio.shape}."
    )
    audio = audio[: spec.sample_length]
    x = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))[None, :, None]
    x = x.permute(0, 2, 1).float()                      # VQVAE.preprocess
    xe = vqvae_encoder_forward(w, x, spec)
    zs_top = bottleneck_encode(w["bottleneck.level_blocks.2.k"], xe)
    z = zs_top.flatten()[None, :]
    if z.shape[-1] < spec.n_ctx:
        raise ValueError("Audio file is not long enough")
    return z


# ---------------------------------------------------------------------------------------------
# Conditioning (jukebox/prior/conditioners.py LabelConditioner / RangeEmbedding)
# ---------------------------------------------------------------------------------------------
def _range_embedding(emb: torch.Tensor, n_time: int, bins: int, rng, pos_start, pos_end=None, clamp=False):
    pos_min, pos_max = rng
    pos_start = pos_start.float()
    if pos_end is not None:
        if clamp:
            pos_end = pos_end.clamp(pos_min, pos_max)
        pos_end = pos_end.float()
    if n_time != 1:
        interpolation = (torch.arange(0, n_time, dtype=torch.float).view(1, n_time) / n_time)
        position = pos_start + (pos_end - pos_start) * interpolation
    else:
        position = pos_start
    normalised_position = (position - pos_min) / (pos_max - pos_min)
    b = (bins * normalised_position).floor().long()
    return emb[b]


def get_cond(w: Dict[str, torch.Tensor], spec: Spec):
    """jukebox/main.py:71-98.  Returns x_cond (1,n_ctx,width), y_cond (1,1,width)."""
    sample_length = (int(spec.cond_seconds * spec.sr) // spec.raw_to_tokens) * spec.raw_to_tokens
    # labeller.get_label: [total_length, offset, sample_length, artist_id, genre_ids(padded -1)]
    y = torch.tensor(
        [[sample_length, 0, spec.n_ctx * spec.raw_to_tokens, 0, 0] + [-1] * (spec.max_bow_genre_size - 1)],
        dtype=torch.long,
    )
    total_length, offset, length, artist, genre = y[:, 0:1], y[:, 1:2], y[:, 2:3], y[:, 3:4], y[:, 4:]
    artist_emb = w["y_emb.artist_emb.emb.weight"][artist]
    mask = (genre >= 0).float().unsqueeze(2)
    genre_emb = (w["y_emb.bow_genre_emb.emb.weight"][genre.clamp(0)] * mask).sum(dim=1, keepdim=True)
    start_emb = genre_emb + artist_emb                                    # (1,1,width)
    start, end = offset, offset + length
    total_length, start, end = total_length.float(), start.float(), end.float()
    sr = spec.sr
    t_ranges = ((spec.min_duration * sr, spec.max_duration * sr), (0.0, spec.max_duration * sr), (0.0, 1.0))
    pos_emb = (
        _range_embedding(w["y_emb.total_length_emb.emb.weight"], 1, spec.t_bins, t_ranges[0], total_length)
        + _range_embedding(w["y_emb.absolute_pos_emb.emb.weight"], spec.n_ctx, spec.t_bins, t_ranges[1], start, end)
        + _range_embedding(
            w["y_emb.relative_pos_emb.emb.weight"], spec.n_ctx, spec.t_bins, t_ranges[2],
            start / total_length, end / total_length, clamp=True,
        )
    )
    x_cond = pos_emb[0, : spec.n_ctx][None, ...]
    y_cond = start_emb[0][None, ...]
    return x_cond, y_cond


# ---------------------------------------------------------------------------------------------
# Top prior (jukebox/prior/autoregressive.py, jukebox/transformer/*)
# ---------------------------------------------------------------------------------------------
def _conv1d_linear(x, wm, b, dtype=torch.float32):
    """transformer/ops.py Conv1D: addmm(b, x.view(-1,n_in), w) with fp16-valued w used as fp32.
    (dtype=float64 is the generator's noise-floor mode, tests/golden/make_jukebox_wide_golden.py: same graph, wider type.)"""
    size_out = (*x.size()[:-1], wm.shape[1])
    return torch.addmm(b.to(dtype), x.reshape(-1, x.size(-1)), wm.to(dtype)).view(*size_out)


def _dense_attn(q, k, v, heads: int, causal: bool, dtype=torch.float32):
    bs, ql, d = q.shape
    kl = k.shape[1]
    hd = d // heads
    qh = q.view(bs, ql, heads, hd).permute(0, 2, 1, 3)
    kh = k.view(bs, kl, heads, hd).permute(0, 2, 3, 1)
    vh = v.view(bs, kl, heads, hd).permute(0, 2, 1, 3)
    scale = 1.0 / math.sqrt(math.sqrt(hd))
    wt = torch.matmul(qh, kh)
    wt.mul_(scale * scale)
    wt = wt.to(dtype)
    if causal:
        mask = torch.ones(ql, kl, dtype=dtype).tril(max(0, kl - ql)).view(1, 1, ql, kl)
        wt = wt * mask + -1e9 * (1 - mask)
    wt = F.softmax(wt, dim=-1)
    a = torch.matmul(wt, vh)
    return a.permute(0, 2, 1, 3).contiguous().view(bs, ql, d)


def factored_attention(q, k, v, attn_func: int, heads: int, block_ctx: int, dtype=torch.float32):
    """factored_attention.py block_attn (1) / transpose_block_attn (2) / prev_block_attn (3)."""
    bs, l, d = v.shape
    if attn_func == 1:
        qq = q.view(bs * l // block_ctx, block_ctx, d)
        kk = k.view(bs * l // block_ctx, block_ctx, d)
        vv = v.view(bs * l // block_ctx, block_ctx, d)
        return _dense_attn(qq, kk, vv, heads, True, dtype).view(bs, l, d)
    if attn_func == 2:
        def tr(x):
            return x.view(bs, l // block_ctx, block_ctx, d).transpose(1, 2).contiguous().view(
                bs * block_ctx, l // block_ctx, d)
        a = _dense_attn(tr(q), tr(k), tr(v), heads, True, dtype)
        return a.view(bs, block_ctx, l // block_ctx, d).transpose(1, 2).contiguous().view(bs, l, d)
    if attn_func == 3:
        qq = q.view(bs * l // block_ctx, block_ctx, d)
        def prev(x):
            return F.pad(x.view(bs, l // block_ctx, block_ctx, d)[:, :-1, :, :], (0, 0, 0, 0, 1, 0)).view(
                bs * l // block_ctx, block_ctx, d)
        return _dense_attn(qq, prev(k), prev(v), heads, False, dtype).view(bs, l, d)
    raise ValueError(attn_func)


def prior_embed(w, z, x_cond, y_cond, spec: Spec):
    """autoregressive.py forward head: x_emb -> roll(+1) -> x[:,0]=y_cond -> + pos_emb + x_cond."""
    N = z.shape[0]
    x = w["prior.x_emb.weight"][z]
    x = torch.cat((x[:, -1:], x[:, :-1]), dim=1)          # roll(x, 1)
    x[:, 0] = y_cond.view(-1, spec.prior_width).expand(N, -1)
    x = x + w["prior.pos_emb.pos_emb"] + x_cond
    return x


def prior_layer(w, x, d: int, spec: Spec, taps: Optional[dict] = None, dtype=torch.float32):
    """ResAttnBlock: a = attn(ln_0(x)); m = mlp(ln_1(x + a)); h = x + a + m (res_scale == 1).
    dtype=float64 evaluates the same graph in double (noise-floor reference of the wide fixture; x must be double too)."""
    p = f"prior.transformer._attn_mods.{d}"
    W = spec.prior_width
    f = lambda t: t.to(dtype)  # noqa: E731
    ln0 = F.layer_norm(f(x), (W,), f(w[f"{p}.ln_0.weight"]), f(w[f"{p}.ln_0.bias"]), 1e-5)
    qkv = _conv1d_linear(ln0, w[f"{p}.attn.c_attn.w"], w[f"{p}.attn.c_attn.b"], dtype)
    q, k, v = qkv.chunk(3, dim=2)
    att = factored_attention(q.contiguous(), k.contiguous(), v.contiguous(), [1, 2, 3][d % 3], spec.heads, spec.block_ctx, dtype)
    a = _conv1d_linear(att, w[f"{p}.attn.c_proj.w"], w[f"{p}.attn.c_proj.b"], dtype)
    xa = x + a
    ln1 = F.layer_norm(f(xa), (W,), f(w[f"{p}.ln_1.weight"]), f(w[f"{p}.ln_1.bias"]), 1e-5)
    fc = _conv1d_linear(ln1, w[f"{p}.mlp.c_fc.w"], w[f"{p}.mlp.c_fc.b"], dtype)
    g = fc * torch.sigmoid(1.702 * fc)
    m = _conv1d_linear(g, w[f"{p}.mlp.c_proj.w"], w[f"{p}.mlp.c_proj.b"], dtype)
    if taps is not None:
        taps.update(ln0=ln0, qkv=qkv, att=att, xa=xa, ln1=ln1, g=g)
    return xa + m


def get_final_activations(z, x_cond, y_cond, w, spec: Spec, depth: Optional[int] = None):
    """jukebox/main.py:101-110: prior.forward(only_encode=True, fp16=False); merged_decoder=True so
    no x_cond is re-added after the transformer, and there is no final LayerNorm."""
    x = z[:, : spec.n_ctx]
    h = prior_embed(w, x, x_cond, y_cond, spec)
    for d in range(spec.prior_depth if depth is None else depth):
        h = prior_layer(w, h, d, spec)
    return h.float()


def windowed_average(acts: torch.Tensor, frame_len: int, ceil_mode=False):
    """jukebox/main.py:113-130."""
    assert acts.ndim == 2, "expected 2d inputs"
    acts = torch.unsqueeze(acts, 0)
    acts = torch.transpose(acts, 1, 2)
    pool = torch.nn.AvgPool1d(frame_len, stride=frame_len, ceil_mode=ceil_mode)
    acts = pool(acts)
    return torch.transpose(acts, 1, 2)


def normalize_audio(audio: np.ndarray) -> np.ndarray:
    """jukebox/main.py:36-45 minus the librosa decode: mono mean + peak normalise."""
    audio = np.asarray(audio, dtype=np.float32)
    if audio.ndim == 1:
        audio = audio[np.newaxis]
    audio = audio.mean(axis=0)
    norm_factor = np.abs(audio).max()
    if norm_factor > 0:
        audio = audio / norm_factor
    return audio.flatten()


def get_acts_from_audio(audio: np.ndarray, w, spec: Spec, meanpool=True, pool_frames_per_second=None,
                        depth: Optional[int] = None):
    """jukebox/main.py:133-173 with the decoded waveform passed in (no librosa offline)."""
    audio = normalize_audio(audio)
    input_audio_len = len(audio)
    latent_audio_len = math.floor(spec.n_ctx * input_audio_len / spec.sample_length)
    if len(audio) < spec.sample_length:
        audio = np.pad(audio, (0, spec.sample_length - len(audio)))
    z = get_z(audio, w, spec)
    x_cond, y_cond = get_cond(w, spec)
    acts = get_final_activations(z, x_cond, y_cond, w, spec, depth=depth)
    acts = acts.squeeze(0).type(torch.float32)
    acts = acts[:latent_audio_len, :]
    if meanpool:
        if not pool_frames_per_second:
            acts = acts.mean(dim=0)
        else:
            acts_rate = spec.n_ctx / (spec.sample_length / spec.sr)
            frame_len = math.floor(acts_rate / pool_frames_per_second)
            acts = windowed_average(acts, frame_len)
            acts = torch.squeeze(acts, 0)
    return acts.cpu().numpy().copy()
