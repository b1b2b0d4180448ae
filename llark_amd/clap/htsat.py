"""HTSAT-base (Swin) audio encoder of CLAP on the HIP library: the alternative audio-encoder plugin of the reference
(``scripts/clap/clap_embeddings.py:63-107``: ``laion_clap.CLAP_Module(enable_fusion=False, amodel="HTSAT-base")`` ->
``model.get_audio_embedding`` -> one 512-d L2-normalised vector per clip, written as ``<id>.npy``).

Everything runs through ``libllark_hip.so`` (no torch math on the data path; torch allocates buffers and holds weights):
GEMMs on gemm.hip, LayerNorm / GELU on mpt.hip, the Swin-specific gathers and window attention on clap.hip.

Precision: ``"fp32"`` keeps fp32-class accuracy on 16-bit MFMA by splitting BOTH operands into bf16 hi + lo planes
(A_hi.W_hi + A_lo.W_hi from the split-mode kernel, plus an A_hi.W_lo correction launch accumulated through the residual
epilogue); ``"bf16"`` rounds weights and activations to bf16 (one MFMA pass, fp32 accumulate and fp32 residual stream).
Weights use the parameter names of ``transformers.ClapAudioModelWithProjection`` (``audio_model.audio_encoder.*``,
``audio_projection.*``); :func:`from_laion_state_dict` renames a laion_clap checkpoint (``audio_branch.*``) to them.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import ops as O


@dataclass
class ClapDims:
    """HTSAT-base as configured by the reference (CLAP_MODEL_CFG, clap_embeddings.py:109-122)."""
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
    def out_width(self) -> int:
        return self.embed_dim * 2 ** (len(self.depths) - 1)


def bicubic_time_taps(frames: int, out_frames: int):
    """4-tap table of ``F.interpolate(mode="bicubic", align_corners=True)`` along one axis (ATen's cubic convolution,
    A = -0.75, clamped border indices), in fp32 like ATen's CPU kernel: returns (idx int32 [out][4], w fp32 [out][4]).
    ``frames == out_frames`` gives the identity table; longer inputs are rejected like the reference model does."""
    if frames > out_frames:
        raise ValueError("the wav size should be less than or equal to the swin input size")
    idx = np.zeros((out_frames, 4), np.int32)
    w = np.zeros((out_frames, 4), np.float32)
    if frames == out_frames:
        idx[:] = np.arange(out_frames, dtype=np.int32)[:, None]
        w[:, 1] = 1.0
        return idx, w
    f32 = np.float32
    A = f32(-0.75)
    scale = f32(frames - 1) / f32(out_frames - 1) if out_frames > 1 else f32(0)
    real = (scale * np.arange(out_frames, dtype=np.float32)).astype(np.float32)
    x0 = np.minimum(np.floor(real).astype(np.int64), frames - 1)
    t = np.clip(real - x0.astype(np.float32), f32(0), f32(1)).astype(np.float32)

    def conv1(x):
        return (((A + f32(2)) * x - (A + f32(3))) * x * x + f32(1)).astype(np.float32)

    def conv2(x):
        return ((((A * x - f32(5) * A) * x) + f32(8) * A) * x - f32(4) * A).astype(np.float32)

    w[:, 0], w[:, 1], w[:, 2], w[:, 3] = conv2(t + f32(1)), conv1(t), conv1(f32(1) - t), conv2(f32(2) - t)
    for j in range(4):
        idx[:, j] = np.clip(x0 + j - 1, 0, frames - 1)
    return idx, w


def from_laion_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """laion_clap checkpoint names (``[module.]audio_branch.*`` / ``audio_projection.{0,2}.*``, fused ``attn.qkv``) -> the
    names this engine loads.  Keys of the text branch and the spectrogram-extractor buffers are dropped."""
    out = {}
    for k, v in sd.items():
        k = k[7:] if k.startswith("module.") else k
        if k.startswith("audio_projection."):
            out[k.replace("audio_projection.0.", "audio_projection.linear1.").replace("audio_projection.2.", "audio_projection.linear2.")] = v
            continue
        if not k.startswith("audio_branch."):
            continue
        r = k[len("audio_branch."):]
        e = "audio_model.audio_encoder."
        if r.startswith("bn0."):
            out[e + "batch_norm." + r[4:]] = v
        elif r.startswith("patch_embed.") or r.startswith("norm."):
            out[e + r] = v
        elif r.startswith("layers."):
            r = (r.replace(".norm1.", ".layernorm_before.").replace(".norm2.", ".layernorm_after.")
                  .replace(".attn.proj.", ".attention.output.dense.").replace(".mlp.fc1.", ".intermediate.dense.")
                  .replace(".mlp.fc2.", ".output.dense.")
                  .replace(".attn.relative_position_bias_table", ".attention.self.relative_position_bias_table"))
            if ".attn.qkv." in r:
                C = v.shape[0] // 3
                for i, nm in enumerate(("query", "key", "value")):
                    out[e + r.replace(".attn.qkv.", f".attention.self.{nm}.")] = v[i * C:(i + 1) * C]
            elif "relative_position_index" not in r and "attn_mask" not in r:
                out[e + r] = v
    return out


def random_state_dict(dims: Optional["ClapDims"] = None, device="cpu", seed: int = 0, std: float = 0.05) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights of the right shapes (benchmarks and smoke runs; there is no checkpoint on disk)."""
    d = dims or ClapDims()
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *shape: torch.randn(*shape, generator=g, device=device) * std
    e = "audio_model.audio_encoder"
    sd = {f"{e}.batch_norm.weight": 1 + r(d.mel_bins), f"{e}.batch_norm.bias": r(d.mel_bins),
          f"{e}.batch_norm.running_mean": r(d.mel_bins) * 40 - 30, f"{e}.batch_norm.running_var": 100 + r(d.mel_bins).abs() * 400,
          f"{e}.patch_embed.proj.weight": r(d.embed_dim, 1, d.patch, d.patch) * 4, f"{e}.patch_embed.proj.bias": r(d.embed_dim),
          f"{e}.patch_embed.norm.weight": 1 + r(d.embed_dim), f"{e}.patch_embed.norm.bias": r(d.embed_dim)}
    C = d.embed_dim
    for s, depth in enumerate(d.depths):
        M = int(C * d.mlp_ratio)
        for b in range(depth):
            p = f"{e}.layers.{s}.blocks.{b}"
            for ln in ("layernorm_before", "layernorm_after"):
                sd[f"{p}.{ln}.weight"], sd[f"{p}.{ln}.bias"] = 1 + r(C), r(C)
            sd[f"{p}.attention.self.relative_position_bias_table"] = r((2 * d.window - 1) ** 2, d.heads[s]) * 4
            for nm, (n, k) in (("attention.self.query", (C, C)), ("attention.self.key", (C, C)), ("attention.self.value", (C, C)),
                               ("attention.output.dense", (C, C)), ("intermediate.dense", (M, C)), ("output.dense", (C, M))):
                sd[f"{p}.{nm}.weight"], sd[f"{p}.{nm}.bias"] = r(n, k), r(n)
        if s < len(d.depths) - 1:
            p = f"{e}.layers.{s}.downsample"
            sd[f"{p}.reduction.weight"], sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"] = r(2 * C, 4 * C), 1 + r(4 * C), r(4 * C)
            C *= 2
    sd[f"{e}.norm.weight"], sd[f"{e}.norm.bias"] = 1 + r(C), r(C)
    sd["audio_projection.linear1.weight"], sd["audio_projection.linear1.bias"] = r(d.proj_dim, C), r(d.proj_dim)
    sd["audio_projection.linear2.weight"], sd["audio_projection.linear2.bias"] = r(d.proj_dim, d.proj_dim), r(d.proj_dim)
    return sd


def algorithmic_flops_per_clip(dims: Optional["ClapDims"] = None) -> Dict[str, float]:
    """2*M*N*K of every matrix product of one clip (fp32-equivalent; extra MFMA passes of the split modes not counted)."""
    d = dims or ClapDims()
    L, C = (d.spec_size // d.patch) ** 2, d.embed_dim
    gemm = 2.0 * L * d.patch * d.patch * C
    attn = 0.0
    for s, depth in enumerate(d.depths):
        M = int(C * d.mlp_ratio)
        gemm += depth * 2.0 * L * (4 * C * C + 2 * C * M)
        attn += depth * 4.0 * L * d.window * d.window * C
        if s < len(d.depths) - 1:
            gemm += 2.0 * (L // 4) * 4 * C * 2 * C
            L, C = L // 4, 2 * C
    gemm += 2.0 * (C * d.proj_dim + d.proj_dim * d.proj_dim)
    return {"gemm": gemm, "attention": attn}


class _Linear:
    """nn.Linear weight [n][k] as bf16 hi (+ lo) K-contiguous planes, bias fp32."""

    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], device, fp32: bool):
        w = w.to(device=device, dtype=torch.float32).contiguous()
        self.n, self.k = w.shape
        self.hi = O.pack_weight16(w, False, torch.bfloat16)
        self.lo = O.pack_weight16((w - self.hi[:, :self.k].float()).contiguous(), False, torch.bfloat16) if fp32 else None
        self.bias = b.to(device=device, dtype=torch.float32).contiguous() if b is not None else None
        # fp32-class product in ONE launch: [A_hi | A_lo | A_hi] . [W_hi | W_hi | W_lo]^T (K-concatenated, non-split kernel)
        self.w3 = torch.cat([self.hi, self.hi, self.lo], dim=1).contiguous() if fp32 else None


class HipClapAudioEncoder:
    """``embed(log_mel)`` = laion_clap's ``get_audio_embedding`` from the log-mel spectrogram on (its ``bn0`` onward)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], dims: Optional[ClapDims] = None, device="cuda:0", precision: str = "fp32"):
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.dims = d = dims or ClapDims()
        self.device = torch.device(device)
        self.fp32 = precision == "fp32"
        if d.window != 8 or d.spec_size % d.mel_bins or (d.spec_size // d.patch) % d.window:
            raise ValueError("HipClapAudioEncoder: window must be 8 and the token map a multiple of it")
        sd = state_dict
        e = "audio_model.audio_encoder"
        f32 = lambda k: sd[k].to(device=self.device, dtype=torch.float32).contiguous()
        lin = lambda p, bias=True: _Linear(sd[p + ".weight"], sd[p + ".bias"] if bias else None, self.device, self.fp32)
        var = f32(f"{e}.batch_norm.running_var")
        self.bn_mean = f32(f"{e}.batch_norm.running_mean")
        self.bn_scale = (f32(f"{e}.batch_norm.weight") / torch.sqrt(var + d.bn_eps)).contiguous()     # load-time constant folding
        self.bn_bias = f32(f"{e}.batch_norm.bias")
        pw = sd[f"{e}.patch_embed.proj.weight"]
        self.patch_proj = _Linear(pw.reshape(pw.shape[0], -1), sd[f"{e}.patch_embed.proj.bias"], self.device, self.fp32)
        self.patch_norm = (f32(f"{e}.patch_embed.norm.weight"), f32(f"{e}.patch_embed.norm.bias"))
        self.stages = []
        for s, depth in enumerate(d.depths):
            blocks = []
            for b in range(depth):
                p = f"{e}.layers.{s}.blocks.{b}"
                a = f"{p}.attention.self"
                qkv_w = torch.cat([sd[f"{a}.{n}.weight"] for n in ("query", "key", "value")], 0)
                qkv_b = torch.cat([sd[f"{a}.{n}.bias"] for n in ("query", "key", "value")], 0)
                blocks.append(dict(
                    ln1=(f32(f"{p}.layernorm_before.weight"), f32(f"{p}.layernorm_before.bias")),
                    ln2=(f32(f"{p}.layernorm_after.weight"), f32(f"{p}.layernorm_after.bias")),
                    qkv=_Linear(qkv_w, qkv_b, self.device, self.fp32), proj=lin(f"{p}.attention.output.dense"),
                    fc1=lin(f"{p}.intermediate.dense"), fc2=lin(f"{p}.output.dense"),
                    bias_table=f32(f"{a}.relative_position_bias_table")))
            down = None
            if s < len(d.depths) - 1:
                p = f"{e}.layers.{s}.downsample"
                down = dict(norm=(f32(f"{p}.norm.weight"), f32(f"{p}.norm.bias")), red=lin(f"{p}.reduction", bias=False))
            self.stages.append((blocks, down))
        self.norm = (f32(f"{e}.norm.weight"), f32(f"{e}.norm.bias"))
        self.proj1, self.proj2 = lin("audio_projection.linear1"), lin("audio_projection.linear2")
        self._taps = {}
        self._arena: Dict[str, torch.Tensor] = {}
        self._arena_version = 0
        self._plans = {}
        self.replay = os.environ.get("LLARK_CLAP_REPLAY", "1") != "0"
        self.kcat = self.fp32 and os.environ.get("LLARK_CLAP_KCAT", "1") != "0"        # one-launch linears (see _Linear.w3)
        self.fuse_gelu = os.environ.get("LLARK_CLAP_FUSE_GELU", "1") != "0"           # exact GELU in the fc1 epilogue

    # ---- persistent workspaces + recorded launch lists ----
    def _ws(self, name: str, shape, dtype) -> torch.Tensor:
        """View of the persistent flat buffer `name` (grown on demand; growth invalidates recorded launch lists)."""
        n = 1
        for v in shape:
            n *= int(v)
        buf = self._arena.get(name)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty(n, dtype=dtype, device=self.device)
            self._arena[name] = buf
            self._arena_version += 1
        return buf[:n].view(*shape)

    def _planes(self, name: str, rows: int, width: int, zero: bool = False):
        hi = self._ws(name + "_hi", (rows, width), torch.bfloat16)
        lo = self._ws(name + "_lo", (rows, width), torch.bfloat16) if self.fp32 else None
        if zero:                                      # pad columns of a K < 32 operand: written once, never touched by the kernels
            hi.zero_()
            if lo is not None:
                lo.zero_()
        return hi, lo

    def _linear(self, a_hi, a_lo, lin: _Linear, c: torch.Tensor, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
        O.gemm16(a_hi, a_lo, lin.hi, lin.bias, lin.n, O.EPI_RESID if resid is not None else O.EPI_F32, c=c, resid=resid)
        if lin.lo is not None:
            O.gemm16(a_hi, None, lin.lo, None, lin.n, O.EPI_RESID, c=c, resid=c)
        return c

    def _time_taps(self, frames: int):
        if frames not in self._taps:
            d = self.dims
            idx, w = bicubic_time_taps(frames, d.spec_size * (d.spec_size // d.mel_bins))
            self._taps[frames] = (torch.from_numpy(idx).to(self.device), torch.from_numpy(w).to(self.device))
        return self._taps[frames]

    # ---- forward ----
    def embed(self, input_features: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """input_features (B, 1, frames, mel) or (B, frames, mel) fp32 log-mel on the GPU -> (B, proj_dim) fp32.
        The first call for a (B, frames) shape runs the layer loop and records its launches; later calls copy the input
        into the persistent buffer and replay the list (``LLARK_CLAP_REPLAY=0`` always runs the loop)."""
        d = self.dims
        x = input_features
        if x.dim() == 4:
            x = x[:, 0]
        if x.dim() != 3 or x.shape[2] != d.mel_bins:
            raise ValueError(f"expected (B, [1,] frames, {d.mel_bins}) log-mel features, got {tuple(input_features.shape)}")
        if not x.is_cuda:
            raise O._lib.LlarkHipError("input_features: tensor must live on the GPU (there is no CPU fallback)")
        B, frames, _ = x.shape
        tap_idx, tap_w = self._time_taps(frames)          # raises for frames > spec * freq_ratio, before anything is launched
        key = (B, frames, bool(normalize))
        plan = self._plans.get(key)
        if plan is not None and plan[0] == self._arena_version and self.replay:
            plan[1].copy_(x)
            plan[2].replay()
            return plan[3].clone()
        x_in = self._ws("x_in", (B, frames, d.mel_bins), torch.float32)
        x_in.copy_(x)
        self._planes("patch", B * (d.spec_size // d.patch) ** 2, O.round_up(d.patch * d.patch, 32), zero=True)
        self._planes("relu", B, O.round_up(self.proj1.n, 32), zero=True)
        self._planes("pool", B, O.round_up(d.out_width, 32), zero=True)
        with O.LaunchList.record() as ll:
            out = self._forward(x_in, tap_idx, tap_w, normalize)
        # a buffer's first use in a forward is its largest (token count x width halves per stage), so nothing recorded
        # above points at a buffer that was regrown later in the same pass; any later growth bumps the version
        self._plans[key] = (self._arena_version, x_in, ll, out)
        return out.clone()

    def _forward(self, x: torch.Tensor, tap_idx, tap_w, normalize: bool) -> torch.Tensor:
        d = self.dims
        B = x.shape[0]
        G = d.spec_size // d.patch
        rows, C = B * G * G, d.embed_dim
        p_hi, p_lo = self._planes("patch", rows, O.round_up(d.patch * d.patch, 32))
        O.clap_patchify(x, self.bn_mean, self.bn_scale, self.bn_bias, tap_idx, tap_w, d.spec_size, d.patch, p_hi, p_lo)
        h = self._linear(p_hi, p_lo, self.patch_proj, self._ws("h0", (rows, C), torch.float32))
        O.layernorm_f32_(h, self.patch_norm[0], self.patch_norm[1], d.ln_eps)
        H = W = G
        for s, (blocks, down) in enumerate(self.stages):
            heads = d.heads[s]
            for b, blk in enumerate(blocks):
                shift = d.window // 2 if (b % 2 == 1 and min(H, W) > d.window) else 0
                M = blk["fc1"].n
                if self.kcat:
                    a3 = self._ws("act3", (rows, 3 * C), torch.bfloat16)
                    a_hi, a_lo, a_dup = a3[:, :C], a3[:, C:2 * C], a3[:, 2 * C:]
                    m3 = self._ws("mlp3", (rows, 3 * M), torch.bfloat16)
                    O.layernorm_bf16_dup(h, blk["ln1"][0], blk["ln1"][1], d.ln_eps, a_hi, a_lo, a_dup)
                    qkv = self._ws("qkv", (rows, 3 * C), torch.float32)
                    O.gemm16(a3, None, blk["qkv"].w3, blk["qkv"].bias, 3 * C, O.EPI_F32, c=qkv)
                    O.clap_window_attn(qkv, B, H, W, C, heads, d.window, shift, blk["bias_table"], a_hi, a_lo, a_dup)
                    O.gemm16(a3, None, blk["proj"].w3, blk["proj"].bias, C, O.EPI_RESID, c=h, resid=h)
                    O.layernorm_bf16_dup(h, blk["ln2"][0], blk["ln2"][1], d.ln_eps, a_hi, a_lo, a_dup)
                    if self.fuse_gelu:
                        O.gemm16_act(a3, None, blk["fc1"].w3, blk["fc1"].bias, M, m3[:, :M], m3[:, M:2 * M], m3[:, 2 * M:], act=2)
                    else:
                        mid = self._ws("mid", (rows, M), torch.float32)
                        O.gemm16(a3, None, blk["fc1"].w3, blk["fc1"].bias, M, O.EPI_F32, c=mid)
                        O.gelu_split_bf16(mid, m3[:, :M], m3[:, M:2 * M])
                        m3[:, 2 * M:].copy_(m3[:, :M])
                    O.gemm16(m3, None, blk["fc2"].w3, blk["fc2"].bias, C, O.EPI_RESID, c=h, resid=h)
                    continue
                a_hi, a_lo = self._planes("act", rows, C)
                O.layernorm_bf16(h, blk["ln1"][0], blk["ln1"][1], d.ln_eps, a_hi, a_lo)
                qkv = self._linear(a_hi, a_lo, blk["qkv"], self._ws("qkv", (rows, 3 * C), torch.float32))
                O.clap_window_attn(qkv, B, H, W, C, heads, d.window, shift, blk["bias_table"], a_hi, a_lo)     # planes reused for the context
                self._linear(a_hi, a_lo, blk["proj"], h, resid=h)
                O.layernorm_bf16(h, blk["ln2"][0], blk["ln2"][1], d.ln_eps, a_hi, a_lo)
                m_hi, m_lo = self._planes("mlp", rows, M)
                if self.fuse_gelu and not self.fp32:
                    O.gemm16_act(a_hi, None, blk["fc1"].hi, blk["fc1"].bias, M, m_hi, act=2)
                else:
                    mid = self._linear(a_hi, a_lo, blk["fc1"], self._ws("mid", (rows, M), torch.float32))
                    O.gelu_split_bf16(mid, m_hi, m_lo)
                self._linear(m_hi, m_lo, blk["fc2"], h, resid=h)
            if down is not None:
                merged = self._ws("qkv", (rows // 4, 4 * C), torch.float32)               # the qkv buffer is free between blocks
                O.clap_patch_merge(h, B, H, W, merged)
                rows, H, W = rows // 4, H // 2, W // 2
                h_new = self._ws(f"h{(s + 1) % 2}", (rows, 2 * C), torch.float32)
                if self.kcat:
                    g3 = self._ws("act3", (rows, 12 * C), torch.bfloat16)
                    O.layernorm_bf16_dup(merged, down["norm"][0], down["norm"][1], d.ln_eps, g3[:, :4 * C], g3[:, 4 * C:8 * C], g3[:, 8 * C:])
                    O.gemm16(g3, None, down["red"].w3, None, 2 * C, O.EPI_F32, c=h_new)
                else:
                    g_hi, g_lo = self._planes("act", rows, 4 * C)
                    O.layernorm_bf16(merged, down["norm"][0], down["norm"][1], d.ln_eps, g_hi, g_lo)
                    self._linear(g_hi, g_lo, down["red"], h_new)
                h, C = h_new, 2 * C
        O.layernorm_f32_(h, self.norm[0], self.norm[1], d.ln_eps)
        pooled = self._ws("pooled", (B, C), torch.float32)
        O.mean_rows_f32(h, B, pooled)
        q_hi, q_lo = self._planes("pool", B, O.round_up(C, 32))
        O.split16_into(pooled, q_hi, q_lo)
        y = self._linear(q_hi, q_lo, self.proj1, self._ws("proj1", (B, self.proj1.n), torch.float32))
        r_hi, r_lo = self._planes("relu", B, O.round_up(self.proj1.n, 32))
        O.relu_split_bf16(y, r_hi, r_lo)
        out = self._linear(r_hi, r_lo, self.proj2, self._ws("out", (B, self.proj2.n), torch.float32))
        if normalize:
            O.l2_normalize_rows_(out)
        return out
