"""Thin, validating Python wrappers over the C-ABI of ``libllark_hip.so``.

PyTorch is used here only as a device-memory container and stream provider: every wrapper takes
device tensors, checks dtype / shape / contiguity, and passes raw pointers plus the current HIP
stream to the library.  There is no fallback: a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check

F16, BF16, F32SRC = 0, 1, 2
EPI_F32, EPI_RESID, EPI_QGELU_SPLIT, EPI_OUT16, EPI_SWIGLU16, EPI_SPLIT16, EPI_SWIGLU_SPLIT, EPI_QGELU_SPLIT8 = 0, 1, 2, 3, 4, 5, 6, 7
LO8_SA = 12      # 2^12: exponent of the activation low plane in lo8 mode (|a - fp16(a)| <= 2^-11 |a|; covers |a| < 2^8 unsaturated)

_DT = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32SRC}


# optional per-kernel HIP-event timing (used by bench.py for the roofline line): a dict
# name -> list[(start_event, end_event, work)] ; events are recorded on the launch stream.
_TIMERS = None


def start_kernel_timing() -> None:
    global _TIMERS
    _TIMERS = {}


def kernel_timing_active() -> bool:
    return _TIMERS is not None


def stop_kernel_timing():
    """Returns {name: (launches, total_ms, total_work)} and disables timing."""
    global _TIMERS
    torch.cuda.synchronize()
    out = {}
    for name, evs in (_TIMERS or {}).items():
        out[name] = (len(evs), sum(a.elapsed_time(b) for a, b, _ in evs), sum(w for _, _, w in evs))
    _TIMERS = None
    return out


class _timed:
    def __init__(self, name, work):
        self.name, self.work = name, work

    def __enter__(self):
        _lib.current_label = (self.name, self.work)
        if _TIMERS is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        _lib.current_label = None
        if _TIMERS is not None:
            self.b.record(torch.cuda.current_stream())
            _TIMERS.setdefault(self.name, []).append((self.a, self.b, self.work))


class LaunchList:
    """A recorded sequence of C-ABI launches (see _lib._RecordingProxy).  ``with LaunchList.record() as ll: forward()``
    runs ``forward`` once and keeps its launches; ``ll.replay()`` re-issues them on the current stream.  Valid only while
    every buffer the recorded calls point at is alive and unmoved: the owner keeps them (persistent workspaces)."""

    def __init__(self):
        self.calls = []
        self.stream = None

    class _Rec:
        def __init__(self, ll):
            self.ll = ll

        def __enter__(self):
            if _lib._recorder is not None:
                raise RuntimeError("LaunchList.record() does not nest")
            self.ll.stream = _stream()
            _lib.set_recorder(self.ll.calls)
            return self.ll

        def __exit__(self, *exc):
            _lib.set_recorder(None)

    @classmethod
    def record(cls):
        return cls._Rec(cls())

    def replay(self) -> None:
        s_now, s_rec = _stream(), self.stream
        timing = _TIMERS is not None
        for fn, name, args, label in self.calls:
            if args and args[-1] == s_rec and s_now != s_rec:
                args = args[:-1] + (s_now,)
            if timing and label is not None:
                with _timed(label[0], label[1]):
                    rc = fn(*args)
            else:
                rc = fn(*args)
            if rc:
                check(rc, name)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=None, contiguous=True) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.LlarkHipError(f"{name}: tensor must live on the GPU (there is no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


# Caller-owned workspaces of the persistent GEMM kernels (include/llark_hip.h: llark_workspace_create): one per
# (device, stream), created on first use, destroyed with the process.  The C library itself keeps no such state.
_workspaces = {}


def workspace() -> Optional[int]:
    key = (torch.cuda.current_device(), _stream())
    ws = _workspaces.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            # creating one allocates: not allowed inside a hipGraph capture.  Without a workspace llark_gemm16_ws simply never
            # picks a persistent tile variant (captured regions are decode steps: M <= 16, skinny kernels, no workspace use).
            return None
        ws = _lib._real_lib().llark_workspace_create()
        if not ws:
            check(-3, "workspace_create")
        _workspaces[key] = ws
    return ws


# Scratch of the stream-K GEMM (llark_gemm16_fragw_sk): fp32 partial tiles + hand-off flags, one per (device, stream), zeroed
# once.  LLARK_STREAMK=0 turns the decomposition off (every product then runs one workgroup per tile).
_sk_scratch = {}


def sk_scratch() -> Optional[torch.Tensor]:
    if os.environ.get("LLARK_STREAMK", "1") == "0":
        return None
    key = (torch.cuda.current_device(), _stream())
    t = _sk_scratch.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        nbytes = _lib._real_lib().llark_gemm16_sk_scratch_bytes()
        if nbytes <= 0:
            check(-3, "gemm16_sk_scratch_bytes")
        t = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device="cuda")
        _sk_scratch[key] = t
    return t


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------------
# VQ-VAE encoder
# ------------------------------------------------------------------------------------------------
def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """torch Conv1d.weight [cout][cin][k] -> kernel layout [k][cin][cout]."""
    cout, cin, k = w.shape
    wp = torch.empty((k, cin, cout), dtype=torch.float32, device=w.device)
    check(_lib.lib().llark_pack_conv_weight(_dev(w, "w", torch.float32), _dev(wp, "wp"), cout, cin, k, _stream()),
          "pack_conv_weight")
    return wp


def conv1d(x: torch.Tensor, wp: torch.Tensor, bias: torch.Tensor, stride: int, pad: int, dil: int = 1,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, cin, tin = x.shape
    k, cin2, cout = wp.shape
    assert cin == cin2, f"conv1d: channel mismatch {cin} vs {cin2}"
    tout = (tin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((n, cout, tout), dtype=torch.float32, device=x.device)
    assert out.shape == (n, cout, tout)
    check(_lib.lib().llark_conv1d_f32(_dev(x, "x", torch.float32), n, cin, tin, _dev(wp, "wp", torch.float32),
                                      _dev(bias, "bias", torch.float32), cout, k, stride, pad, dil,
                                      _dev(out, "out", torch.float32), tout, _stream()), "conv1d")
    return out


def resblock(x: torch.Tensor, w1p, b1, w2p, b2, dil: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, c, t = x.shape
    if out is None:
        out = torch.empty_like(x)
    assert out.shape == x.shape and out.data_ptr() != x.data_ptr(), "resblock is out-of-place (halo reads)"
    check(_lib.lib().llark_resblock_f32(_dev(x, "x", torch.float32), n, c, t, _dev(w1p, "w1p", torch.float32),
                                        _dev(b1, "b1", torch.float32), _dev(w2p, "w2p", torch.float32),
                                        _dev(b2, "b2", torch.float32), dil, _dev(out, "out", torch.float32), _stream()),
          "resblock")
    return out


def vqvae_weight_exponent(w: torch.Tensor) -> int:
    """exp2 with max|w| * 2^exp2 in (2^13, 2^14]: the power-of-two scale a weight tensor is fragment-packed with so that its
    fp16 low plane stays in the normal range (csrc/vqvae_fused.hip)."""
    import math

    amax = float(w.detach().abs().max().float())
    if not (amax > 0.0) or amax != amax:
        return 0
    return max(-40, min(40, 14 - math.ceil(math.log2(amax))))


def vqvae_pack_frag16(w: torch.Tensor, perm1x1: bool = False, exp2: int = 0):
    """torch Conv1d.weight fp32 [cout][cin][k] -> (hi, lo) fp16 MFMA A fragments of w 2^exp2 for llark_vqvae_stage_f16x2
    (include/llark_hip.h: llark_vqvae_pack_frag16)."""
    cout, cin, k = w.shape
    assert cout % 32 == 0 and cin % 16 == 0
    numel = (cout // 32) * (k * cin // 16) * 512
    hi = torch.empty((numel,), dtype=torch.float16, device=w.device)
    lo = torch.empty_like(hi)
    check(_lib.lib().llark_vqvae_pack_frag16(_dev(w, "w", torch.float32), cout, cin, k, int(perm1x1), int(exp2), _dev(hi, "hi"), _dev(lo, "lo"),
                                             _stream()), "vqvae_pack_frag16")
    return hi, lo


def vqvae_stage(x, n: int, cin: int, tin: int, st: dict, out_planes=None, out_f32: Optional[torch.Tensor] = None) -> None:
    """One fused stage of the level encoder (include/llark_hip.h: llark_vqvae_stage_f16x2).  x: audio fp32 (n, tin) when cin == 1,
    else the (hi, lo) fp16 planes [n][tin][cin] of the previous stage; st: the stage's packed weights (jukebox/vqvae.py)."""
    import ctypes

    if cin == 1:
        audio, in_hi, in_lo = _dev(x, "audio", torch.float32), None, None
        assert x.numel() >= n * tin
    else:
        audio, in_hi, in_lo = None, _dev(x[0], "in_hi", torch.float16), _dev(x[1], "in_lo", torch.float16)
        assert x[0].numel() >= n * tin * cin and x[1].numel() >= n * tin * cin
    c = 64 if st["wo_hi"] is not None else 32
    need = n * (tin // 2) * c
    if out_planes is not None:
        assert out_planes[0].numel() >= need and out_planes[1].numel() >= need
    if out_f32 is not None:
        assert out_f32.numel() >= need
    dil = (ctypes.c_int * len(st["dil"]))(*st["dil"])
    wexp = (ctypes.c_int * len(st["wexp"]))(*st["wexp"])
    assert len(st["wexp"]) == 2 + 2 * len(st["dil"])
    opt = lambda t, name, dt: _dev(t, name, dt) if t is not None else None      # noqa: E731
    check(_lib.lib().llark_vqvae_stage_f16x2(
        audio, in_hi, in_lo, n, cin, tin, opt(st["w0f"], "w0f", torch.float32), opt(st["w0_hi"], "w0_hi", torch.float16),
        opt(st["w0_lo"], "w0_lo", torch.float16), _dev(st["b0"], "b0", torch.float32), _dev(st["wr_hi"], "wr_hi", torch.float16),
        _dev(st["wr_lo"], "wr_lo", torch.float16), _dev(st["br"], "br", torch.float32), len(st["dil"]),
        ctypes.cast(dil, ctypes.c_void_p), opt(st["wo_hi"], "wo_hi", torch.float16), opt(st["wo_lo"], "wo_lo", torch.float16),
        opt(st["bo"], "bo", torch.float32), ctypes.cast(wexp, ctypes.c_void_p), out_planes[0].data_ptr() if out_planes is not None else None,
        out_planes[1].data_ptr() if out_planes is not None else None, out_f32.data_ptr() if out_f32 is not None else None, _stream()),
        "vqvae_stage")


def codebook_norms(k: torch.Tensor) -> torch.Tensor:
    bins, emb = k.shape
    kk = torch.empty((bins,), dtype=torch.float32, device=k.device)
    check(_lib.lib().llark_codebook_norms_f32(_dev(k, "k", torch.float32), bins, emb, _dev(kk, "kk"), _stream()),
          "codebook_norms")
    return kk


def codebook_argmin(x: torch.Tensor, k: torch.Tensor, kk: torch.Tensor, want_dist: bool = False):
    n, emb, t = x.shape
    bins = k.shape[0]
    codes = torch.empty((n, t), dtype=torch.int64, device=x.device)
    dist = torch.empty((n, t), dtype=torch.float32, device=x.device) if want_dist else None
    check(_lib.lib().llark_codebook_argmin(_dev(x, "x", torch.float32), n, emb, t, _dev(k, "k", torch.float32),
                                           _dev(kk, "kk", torch.float32), bins, _dev(codes, "codes"),
                                           dist.data_ptr() if dist is not None else None, _stream()), "codebook_argmin")
    return (codes, dist) if want_dist else codes


def codebook_argmin_tie(x: torch.Tensor, k: torch.Tensor, kk: torch.Tensor, tie_a: float, tie_b: float, flag_count: torch.Tensor,
                        flag_list: torch.Tensor, codes: Optional[torch.Tensor] = None) -> torch.Tensor:
    """codebook_argmin + the near-tie certificate: token ids (n_index * t + token) whose best / second-best gap is below
    |x| (tie_a sqrt(d_best) + tie_b |x|) are appended to flag_list (int32, device); flag_count (int32[1], device) is zeroed
    by the call and counts every flagged token, also past the list's capacity."""
    n, emb, t = x.shape
    if codes is None:
        codes = torch.empty((n, t), dtype=torch.int64, device=x.device)
    check(_lib.lib().llark_codebook_argmin_tie(_dev(x, "x", torch.float32), n, emb, t, _dev(k, "k", torch.float32), _dev(kk, "kk", torch.float32),
                                               k.shape[0], _dev(codes, "codes", torch.int64), float(tie_a), float(tie_b),
                                               _dev(flag_count, "flag_count", torch.int32), _dev(flag_list, "flag_list", torch.int32),
                                               flag_list.numel(), _stream()), "codebook_argmin_tie")
    return codes


def vqvae_fix_near_ties(plan, audio: torch.Tensor, raw_to_tokens: int, flag_list: torch.Tensor, count: int, halo_tokens: int, win_tokens: int,
                        win: torch.Tensor, col: torch.Tensor, buf0: torch.Tensor, buf1: torch.Tensor, k: torch.Tensor, kk: torch.Tensor,
                        codes: torch.Tensor) -> None:
    """Exact re-evaluation of flag_list[:count] on receptive-field windows; patches `codes` (n, t_tok) in place."""
    n, t = audio.shape
    assert win.numel() >= count * win_tokens * raw_to_tokens and col.numel() >= count and buf0.numel() == buf1.numel()
    check(_lib.lib().llark_vqvae_fix_near_ties(plan, _dev(audio, "audio", torch.float32), n, t, raw_to_tokens, _dev(flag_list, "flag_list", torch.int32),
                                               int(count), int(halo_tokens), int(win_tokens), _dev(win, "win", torch.float32),
                                               _dev(col, "col", torch.int32), _dev(buf0, "buf0", torch.float32), _dev(buf1, "buf1", torch.float32),
                                               buf0.numel(), _dev(k, "k", torch.float32), _dev(kk, "kk", torch.float32), k.shape[0],
                                               _dev(codes, "codes", torch.int64), _stream()), "vqvae_fix_near_ties")


# ------------------------------------------------------------------------------------------------
# prior
# ------------------------------------------------------------------------------------------------
def prior_embed(z, x_emb, pos_emb, x_cond, y_cond, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, t = z.shape
    bins, width = x_emb.shape
    assert pos_emb.shape == (t, width) and x_cond.numel() == t * width and y_cond.numel() == width
    if out is None:
        out = torch.empty((n, t, width), dtype=torch.float32, device=z.device)
    check(_lib.lib().llark_prior_embed(_dev(z, "z", torch.int64), n, t, width, bins, _dev(x_emb, "x_emb", torch.float32),
                                       _dev(pos_emb, "pos_emb", torch.float32), _dev(x_cond, "x_cond", torch.float32),
                                       _dev(y_cond, "y_cond", torch.float32), _dev(out, "h", torch.float32), _stream()),
          "prior_embed")
    return out


def layernorm_split(x: torch.Tensor, gamma, beta, eps: float, out_hi: torch.Tensor, out_lo: torch.Tensor) -> None:
    """x [rows][width] fp32 -> fp16 hi/lo planes [rows][ldo]."""
    rows, width = x.shape
    assert out_hi.shape == out_lo.shape and out_hi.shape[0] == rows and out_hi.shape[1] >= width
    check(_lib.lib().llark_layernorm_split_f16(_dev(x, "x", torch.float32), x.stride(0), rows, width,
                                               _dev(gamma, "gamma", torch.float32), _dev(beta, "beta", torch.float32),
                                               float(eps), _dev(out_hi, "out_hi", torch.float16),
                                               _dev(out_lo, "out_lo", torch.float16), out_hi.stride(0), _stream()),
          "layernorm_split")


def layernorm_split_lo8(x: torch.Tensor, gamma, beta, eps: float, out_hi: torch.Tensor, out_lo8: torch.Tensor, sa: int = LO8_SA) -> None:
    """x [rows][width] fp32 -> fp16 hi plane [rows][ldo] + E4M3 low plane [rows][ldo8] (uint8, MFMA slot order)."""
    rows, width = x.shape
    assert out_hi.shape[0] == rows and out_lo8.shape[0] == rows and out_hi.shape[1] >= width and out_lo8.dtype == torch.uint8
    check(_lib.lib().llark_layernorm_split_lo8(_dev(x, "x", torch.float32), x.stride(0), rows, width,
                                               _dev(gamma, "gamma", torch.float32), _dev(beta, "beta", torch.float32),
                                               float(eps), _dev(out_hi, "out_hi", torch.float16), out_hi.stride(0),
                                               _dev(out_lo8, "out_lo8", torch.uint8), out_lo8.stride(0), int(sa), _stream()),
          "layernorm_split_lo8")


def lo8_decode(lo8: torch.Tensor, width: int, sa: int = LO8_SA) -> torch.Tensor:
    """Host-side view of an E4M3 low plane (tests / taps): undo the slot order and the 2^sa scale -> fp32 [rows][width]."""
    rows, ld = lo8.shape
    k = torch.arange(ld, device=lo8.device)
    r = k & 63
    pos = (k & ~63) + (((r >> 3) & 1) << 5) + ((r >> 4) << 3) + (r & 7)
    b = lo8[:, pos][:, :width].to(torch.int32)
    e, mnt = (b >> 3) & 15, (b & 7).float()
    mag = torch.where(e == 0, mnt * 2.0 ** -9, (1.0 + mnt * 0.125) * torch.exp2(e.float() - 7.0))     # E4M3: bias 7, subnormal step 2^-9
    return torch.where((b & 128) != 0, -mag, mag) * (2.0 ** -sa)


def prior_attn(qkv: torch.Tensor, n: int, t: int, n_state: int, heads: int, blocks: int, pattern: int,
               out_hi: torch.Tensor, out_lo: torch.Tensor, lo8_sa: Optional[int] = None) -> None:
    assert qkv.shape[0] == n * t and qkv.shape[1] >= 3 * n_state
    if out_lo.dtype == torch.uint8:
        check(_lib.lib().llark_prior_attn_lo8(_dev(qkv, "qkv", torch.float32), qkv.stride(0), n, t, n_state, heads, blocks,
                                              pattern, _dev(out_hi, "out_hi", torch.float16), out_hi.stride(0),
                                              _dev(out_lo, "out_lo8", torch.uint8), out_lo.stride(0),
                                              LO8_SA if lo8_sa is None else int(lo8_sa), _stream()), "prior_attn_lo8")
        return
    check(_lib.lib().llark_prior_attn(_dev(qkv, "qkv", torch.float32), qkv.stride(0), n, t, n_state, heads, blocks,
                                      pattern, _dev(out_hi, "out_hi", torch.float16), _dev(out_lo, "out_lo", torch.float16),
                                      out_hi.stride(0), _stream()), "prior_attn")


def pool_window(h: torch.Tensor, frame_len: int, frames: int) -> torch.Tensor:
    n, t, width = h.shape
    out = torch.empty((n, frames, width), dtype=torch.float32, device=h.device)
    check(_lib.lib().llark_pool_window(_dev(h, "h", torch.float32), n, t, width, frame_len, _dev(out, "out"), frames,
                                       _stream()), "pool_window")
    return out


def pool_mean(h: torch.Tensor, lens: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, t, width = h.shape
    out = torch.empty((n, width), dtype=torch.float32, device=h.device)
    lp = _dev(lens, "lens", torch.int32) if lens is not None else None
    check(_lib.lib().llark_pool_mean(_dev(h, "h", torch.float32), n, t, width, lp, _dev(out, "out"), _stream()),
          "pool_mean")
    return out


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
def pack_weight16(w: torch.Tensor, transpose: bool, dst_dtype: torch.dtype, kmult: int = 32) -> torch.Tensor:
    """Returns wt [n][kp] (K-contiguous, zero padded to a multiple of ``kmult``).

    transpose=True : w is [k][n] (upstream Conv1D.w);  transpose=False: w is [n][k] (nn.Linear.weight).
    """
    if transpose:
        k, n = w.shape
    else:
        n, k = w.shape
    kp = round_up(k, kmult)
    wt = torch.empty((n, kp), dtype=dst_dtype, device=w.device)
    check(_lib.lib().llark_pack_weight16(_dev(w, "w"), _DT[w.dtype], int(transpose), k, n, _dev(wt, "wt"), _DT[dst_dtype],
                                         kp, _stream()), "pack_weight16")
    return wt


def split16(x: torch.Tensor, dtype: torch.dtype, want_lo: bool = True, kmult: int = 32):
    rows, width = x.shape
    ldo = round_up(width, kmult)
    hi = torch.empty((rows, ldo), dtype=dtype, device=x.device)
    lo = torch.empty((rows, ldo), dtype=dtype, device=x.device) if want_lo else None
    check(_lib.lib().llark_split16(_DT[dtype], _dev(x, "x", torch.float32), x.stride(0), rows, width, _dev(hi, "hi"),
                                   lo.data_ptr() if lo is not None else None, ldo, _stream()), "split16")
    return hi, lo


# Fragment-major weight copies for the B-direct kernel live as an attribute ON the row-major tensor object they
# mirror (same lifetime, no address-keyed registry that could go stale).  Attached explicitly by the inference
# engines; a training engine whose weights change in place must not attach.
# From 129 rows (more than one 128-row tile) the B-direct kernels win: with the uniform K split a single clip's prefill (M = 371)
# runs its four products in 0.47 ms per layer against 0.92 ms through the LDS-staged tiles (bench.py --stages llama --batch 1:
# 32.9 -> 19.1 ms split, 18.6 -> 13.2 ms bf16; profiles/r02_streamk.txt).
GEMV_DMA = os.environ.get("LLARK_GEMV_DMA", "1") == "1"      # decode-step Linear (m <= 4, bf16) through the LDS-DMA weight-streaming kernel
# ... for weights of at least this many bytes: the streaming kernel runs at 5.0 TB/s + 3.7 us per launch, the MFMA skinny kernel at
# 4.3 TB/s + 0.4 us (Llama-2-7B o_proj, 33.5 MB: 10.0 vs 8.1 us; qkv, 100 MB: 21.1 vs 22.9; gate_up, 180 MB: 35.1 vs 41.9)
GEMV_DMA_MIN_BYTES = 64 * 1000 * 1000
FRAG_MIN_ROWS = int(os.environ.get("LLARK_FRAG_MIN_ROWS", "129"))


def attach_frag(wt: torch.Tensor, n: int) -> None:
    """Pack and remember the fragment-major copy of `wt` ([>=n][kp], kp % 64 == 0); gemm16 then streams it
    L2 -> VGPR for large-M products.  No-op when the shape does not qualify."""
    if wt.shape[1] % 64 or wt.stride(1) != 1 or os.environ.get("LLARK_FRAG", "1") == "0":
        return
    wt._llark_frag = (pack_weight16_frag(wt, n), n, wt.shape[1])


def detach_frag(wt: torch.Tensor) -> None:
    if hasattr(wt, "_llark_frag"):
        del wt._llark_frag


def _gemv_dma_takes(split: bool, m: int, kp: int) -> bool:
    """The shapes csrc/gemv_dma.hip takes (gemv_shape_ok there): activation fragments in registers or within 48 KiB of LDS."""
    if m < 1 or m > 4 or kp % 8 or kp > 12288:
        return False
    if kp <= 4096 and m <= 2:
        return True
    mm = m if m <= 2 else 4
    return mm * (2 if split else 1) * ((kp + 4095) // 4096) * 8192 <= 48 * 1024


def gemm16(a_hi: torch.Tensor, a_lo: Optional[torch.Tensor], wt: torch.Tensor, bias: Optional[torch.Tensor], n: int,
           epilogue: int, c: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
           out_hi: Optional[torch.Tensor] = None, out_lo: Optional[torch.Tensor] = None, m: Optional[int] = None,
           variant: int = -1) -> None:
    """C[m,n] = (a_hi [+ a_lo]) . wt^T (+bias) with a fused epilogue; see include/llark_hip.h."""
    dtype = a_hi.dtype
    assert dtype in (torch.float16, torch.bfloat16) and wt.dtype == dtype
    m = a_hi.shape[0] if m is None else m
    kp = wt.shape[1]
    assert a_hi.shape[1] >= kp and wt.shape[0] >= n, f"gemm16: A has {a_hi.shape[1]} cols, wt {tuple(wt.shape)}, n={n}"
    name = ("gemm_split_" if a_lo is not None else "gemm_") + ("f16" if dtype == torch.float16 else "bf16") + ("_skinny" if m <= 16 else "")
    if epilogue == EPI_SWIGLU_SPLIT and out_lo is None:
        raise ValueError("gemm16: EPI_SWIGLU_SPLIT needs out_lo")
    if (variant < 0 and GEMV_DMA and dtype == torch.bfloat16 and n * kp * 2 >= GEMV_DMA_MIN_BYTES and _gemv_dma_takes(a_lo is not None, m, kp)
            and epilogue in (EPI_F32, EPI_RESID, EPI_SWIGLU16, EPI_SWIGLU_SPLIT) and a_hi.stride(0) % 8 == 0 and wt.stride(0) % 8 == 0):
        # decode step: the weight-streaming LDS-DMA kernel (csrc/gemv_dma.hip)
        with _timed(name, 2.0 * m * n * kp):
            check(_lib.lib().llark_gemv16_dma(
                int(a_lo is not None), epilogue, _dev(a_hi, "a_hi"), _dev(a_lo, "a_lo", dtype) if a_lo is not None else None, a_hi.stride(0),
                _dev(wt, "wt"), wt.stride(0), _dev(bias, "bias", torch.float32) if bias is not None else None, m, n, kp,
                _dev(c, "c", torch.float32) if c is not None else None, c.stride(0) if c is not None else 0,
                _dev(resid, "resid", torch.float32) if resid is not None else None, resid.stride(0) if resid is not None else 0,
                _dev(out_hi, "out_hi", dtype) if out_hi is not None else None, _dev(out_lo, "out_lo", dtype) if out_lo is not None else None,
                out_hi.stride(0) if out_hi is not None else 0, _stream()), "gemv16_dma")
        return
    if variant < 0 and m >= FRAG_MIN_ROWS:
        fr = getattr(wt, "_llark_frag", None)
        if fr is not None and fr[1] == n and fr[2] == kp:
            return gemm16_fragw(a_hi, a_lo, fr[0], bias, n, kp, epilogue, c=c, resid=resid, out_hi=out_hi, out_lo=out_lo, m=m)
    with _timed(name, 2.0 * m * n * kp):
      check(_lib.lib().llark_gemm16_ws(
        variant, _DT[dtype], int(a_lo is not None), epilogue, _dev(a_hi, "a_hi"), _dev(a_lo, "a_lo", dtype) if a_lo is not None else None,
        a_hi.stride(0), _dev(wt, "wt"), wt.stride(0), _dev(bias, "bias", torch.float32) if bias is not None else None,
        m, n, kp, _dev(c, "c", torch.float32) if c is not None else None, c.stride(0) if c is not None else 0,
        _dev(resid, "resid", torch.float32) if resid is not None else None, resid.stride(0) if resid is not None else 0,
        _dev(out_hi, "out_hi", dtype) if out_hi is not None else None,
        _dev(out_lo, "out_lo", dtype) if out_lo is not None else None,
        out_hi.stride(0) if out_hi is not None else 0, workspace(), _stream()), "gemm16")


def gemm16_ln_takes(m: int, n: int, kp: int) -> bool:
    """True when llark_gemm16_ln (LayerNorm folded into the 256x256 tile's epilogues) takes the shape."""
    return bool(_lib.lib().llark_gemm16_ln_takes(int(m), int(n), int(kp)))


def gemm16_ln(a_hi: torch.Tensor, a_lo: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor], n: int, epilogue: int,
              ln_vec: torch.Tensor, ln_stat: Optional[torch.Tensor] = None, ln_part: Optional[torch.Tensor] = None,
              c: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, out_hi: Optional[torch.Tensor] = None,
              out_lo: Optional[torch.Tensor] = None, m: Optional[int] = None, ln_pred: Optional[torch.Tensor] = None) -> None:
    """The split product with a LayerNorm folded into its epilogue (include/llark_hip.h, llark_gemm16_ln / llark_gemm16_ln_p):
    ``ln_stat`` given = consumer (the operand planes hold x . gamma; the epilogue applies the row's mean / rstd), ``ln_part`` given =
    producer (EPI_RESID; also writes the planes of c . ln_vec and the per-slice sums the next LayerNorm's statistics come from);
    ``ln_pred`` [m][2] (producer only) = the rows' predicted (shift, power-of-two scale): planes of ((c - shift) scale) . ln_vec,
    sums of (c - shift) -- to be reduced by :func:`ln_stats_finalize` with the same ``pred``."""
    dtype = a_hi.dtype
    assert dtype in (torch.float16, torch.bfloat16) and wt.dtype == dtype and a_lo is not None
    m = a_hi.shape[0] if m is None else m
    kp = wt.shape[1]
    assert a_hi.shape[1] >= kp and wt.shape[0] >= n and ln_vec.numel() >= n
    if ln_part is not None:
        assert ln_part.numel() >= m * 2 * ((n + 255) // 256) * 2, "gemm16_ln: ln_part is [m][2 * ceil(n / 256)][2]"
    if ln_stat is not None:
        assert ln_stat.numel() >= 2 * m
    if ln_pred is not None:
        assert ln_part is not None and ln_pred.numel() >= 2 * m and ln_pred.is_contiguous()
    name = "gemm_split_" + ("f16" if dtype == torch.float16 else "bf16")
    with _timed(name, 2.0 * m * n * kp):
        check(_lib.lib().llark_gemm16_ln_p(
            _DT[dtype], epilogue, _dev(a_hi, "a_hi"), _dev(a_lo, "a_lo", dtype), a_hi.stride(0), _dev(wt, "wt"), wt.stride(0),
            _dev(bias, "bias", torch.float32) if bias is not None else None, m, n, kp,
            _dev(c, "c", torch.float32) if c is not None else None, c.stride(0) if c is not None else 0,
            _dev(resid, "resid", torch.float32) if resid is not None else None, resid.stride(0) if resid is not None else 0,
            _dev(out_hi, "out_hi", dtype) if out_hi is not None else None, _dev(out_lo, "out_lo", dtype) if out_lo is not None else None,
            out_hi.stride(0) if out_hi is not None else 0,
            _dev(ln_stat, "ln_stat", torch.float32) if ln_stat is not None else None, _dev(ln_vec, "ln_vec", torch.float32),
            _dev(ln_part, "ln_part", torch.float32) if ln_part is not None else None,
            _dev(ln_pred, "ln_pred", torch.float32) if ln_pred is not None else None, workspace(), _stream()), "gemm16_ln")


def gemm16_lnp_fragw(a_hi: torch.Tensor, a_lo: torch.Tensor, wfrag: torch.Tensor, bias: Optional[torch.Tensor], n: int, kp: int,
                     ln_vec: torch.Tensor, ln_part: torch.Tensor, c: torch.Tensor, resid: torch.Tensor, out_hi: torch.Tensor,
                     out_lo: torch.Tensor, m: Optional[int] = None, ln_pred: Optional[torch.Tensor] = None) -> int:
    """The producer role of :func:`gemm16_ln` with fragment-major weights on 128x256 tiles, two workgroups to a CU
    (include/llark_hip.h, llark_gemm16_lnp_fragw).  Returns the number of 64-column slices ``ln_part`` [m][nparts][2] now holds:
    pass it to :func:`ln_stats_finalize`."""
    dtype = a_hi.dtype
    assert dtype in (torch.float16, torch.bfloat16) and wfrag.dtype == dtype and a_hi.shape[1] >= kp
    assert wfrag.numel() == round_up(n, 32) * kp and ln_vec.numel() >= n
    m = a_hi.shape[0] if m is None else m
    nparts = (n + 63) // 64
    assert ln_part.numel() >= m * nparts * 2, "gemm16_lnp_fragw: ln_part is [m][ceil(n / 64)][2]"
    if ln_pred is not None:
        assert ln_pred.numel() >= 2 * m and ln_pred.is_contiguous()
    name = "gemm_split_" + ("f16" if dtype == torch.float16 else "bf16")
    with _timed(name, 2.0 * m * n * kp):
        check(_lib.lib().llark_gemm16_lnp_fragw(
            _DT[dtype], _dev(a_hi, "a_hi"), _dev(a_lo, "a_lo", dtype), a_hi.stride(0), _dev(wfrag, "wfrag"),
            _dev(bias, "bias", torch.float32) if bias is not None else None, m, n, kp, _dev(c, "c", torch.float32), c.stride(0),
            _dev(resid, "resid", torch.float32), resid.stride(0), _dev(out_hi, "out_hi", dtype), _dev(out_lo, "out_lo", dtype),
            out_hi.stride(0), _dev(ln_vec, "ln_vec", torch.float32), _dev(ln_part, "ln_part", torch.float32),
            _dev(ln_pred, "ln_pred", torch.float32) if ln_pred is not None else None, _stream()), "gemm16_lnp_fragw")
    return nparts


def ln_stats_finalize(part: torch.Tensor, rows: int, nparts: int, width: int, eps: float, stat: torch.Tensor,
                      pred: Optional[torch.Tensor] = None) -> None:
    """part [rows][nparts][2] (sum, sum of squares per column slice) -> stat [rows][2] (mean, rstd).  With ``pred`` [rows][2] (the
    (shift, scale) the producer was given): stat = ((mean - shift) scale, rstd / scale) -- what the consumer role needs for planes of
    ((x - shift) scale) gamma -- and pred is replaced by (mean, nearest power of two of rstd) for the next LayerNorm of these rows."""
    assert part.numel() >= rows * nparts * 2 and stat.numel() >= rows * 2
    if pred is not None:
        assert pred.numel() >= rows * 2 and pred.is_contiguous()
        check(_lib.lib().llark_ln_stats_finalize_p(_dev(part, "part", torch.float32), rows, nparts, width, float(eps),
                                                   _dev(stat, "stat", torch.float32), _dev(pred, "pred", torch.float32), _stream()), "ln_stats_finalize_p")
        return
    check(_lib.lib().llark_ln_stats_finalize(_dev(part, "part", torch.float32), rows, nparts, width, float(eps),
                                             _dev(stat, "stat", torch.float32), _stream()), "ln_stats_finalize")


def ln_row_pred(x: torch.Tensor, eps: float, pred: torch.Tensor) -> None:
    """pred [rows][2] = (mean, nearest power of two of 1 / sqrt(var + eps)) of the rows of x (fp32 [rows][width]): the first prediction
    of a forward for the producers of :func:`gemm16_ln` (ln_pred)."""
    rows, width = x.shape
    assert pred.numel() >= 2 * rows and pred.is_contiguous()
    check(_lib.lib().llark_ln_row_pred(_dev(x, "x", torch.float32), x.stride(0), rows, width, float(eps), _dev(pred, "pred", torch.float32),
                                       _stream()), "ln_row_pred")


def gemm16_t(a: torch.Tensor, wt: torch.Tensor, m: int, n: int, kp: int, trans_a: bool, trans_b: bool, c: torch.Tensor,
             accumulate: bool = False, sumsq: Optional[torch.Tensor] = None, variant: int = -1) -> None:
    """c[m][n] (= | +=) sum_k A(m, k) W(n, k) with operands that may be stored contraction-major (csrc/gemm_tn.hip;
    include/llark_hip.h: llark_gemm16_t): ``trans_a`` -> a is [kp][>= m], ``trans_b`` -> wt is [kp][>= n].
    ``sumsq`` (device double): += the sum of squares of every value written to c (llark_gemm16_t_sumsq)."""
    dtype = a.dtype
    assert dtype in (torch.float16, torch.bfloat16) and wt.dtype == dtype and a.stride(-1) == 1 and wt.stride(-1) == 1
    args = (_DT[dtype], EPI_RESID if accumulate else EPI_F32, int(trans_a), int(trans_b), _dev(a, "a", contiguous=False), a.stride(0),
            _dev(wt, "wt", contiguous=False), wt.stride(0), m, n, kp, _dev(c, "c", torch.float32, contiguous=False), c.stride(0),
            _dev(c, "c", torch.float32, contiguous=False) if accumulate else None, c.stride(0))
    with _timed("gemm_f16" if dtype == torch.float16 else "gemm_bf16", 2.0 * m * n * kp):
        if variant >= 0:                                   # explicit tile / pipeline variant (llark_gemm16_t_ex): benchmarks, A/B tests
            check(_lib.lib().llark_gemm16_t_ex(int(variant), *args, _dev(sumsq, "sumsq", torch.float64) if sumsq is not None else None,
                                               _stream()), "gemm16_t_ex")
        elif sumsq is None:
            check(_lib.lib().llark_gemm16_t(*args, _stream()), "gemm16_t")
        else:
            check(_lib.lib().llark_gemm16_t_sumsq(*args, _dev(sumsq, "sumsq", torch.float64), _stream()), "gemm16_t_sumsq")


def lo8_weight_exponent(w: torch.Tensor) -> int:
    """sw with max|W| * 2^sw <= 448 (the E4M3 maximum): the per-matrix scale of the in-kernel fp8 weight plane."""
    amax = float(w.detach().abs().max().float())
    if not (amax > 0.0) or amax != amax:
        return 0
    import math

    return max(-40, min(40, math.floor(math.log2(448.0 / amax))))


def pack_weight_lo8(wt: torch.Tensor, sw: int) -> torch.Tensor:
    """wt fp16 [n][kp] -> uint8 [n][kp] = e4m3(wt * 2^sw) in MFMA slot order: the staged B operand of the low-plane product."""
    n, kp = wt.shape
    assert wt.dtype == torch.float16 and kp % 64 == 0
    out = torch.empty((n, kp), dtype=torch.uint8, device=wt.device)
    check(_lib.lib().llark_pack_weight_lo8(_dev(wt, "wt"), wt.stride(0), n, kp, int(sw), _dev(out, "out"), out.stride(0), _stream()),
          "pack_weight_lo8")
    return out


def gemm16_lo8(a_hi: torch.Tensor, a_lo8: torch.Tensor, wt: torch.Tensor, sw: int, bias: Optional[torch.Tensor], n: int, epilogue: int,
               c: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, out_hi: Optional[torch.Tensor] = None,
               out_lo8: Optional[torch.Tensor] = None, m: Optional[int] = None, sa: int = LO8_SA, w8: Optional[torch.Tensor] = None) -> None:
    """C[m,n] = a_hi . wt^T + 2^-(sa+sw) a_lo8 . fp8(wt 2^sw)^T (+bias): the prior's split GEMM with an E4M3 low plane
    (include/llark_hip.h: llark_gemm16_lo8; csrc/gemm256_lo8n.hip).  ``w8`` = pack_weight_lo8(wt, sw), required.
    The kernel addresses its operands with 32-bit byte offsets: a product whose A plane reaches 2 GiB (about 28 clips of the
    5b prior) is issued as several launches over row ranges -- rows are independent, so the result is bit-identical."""
    assert a_hi.dtype == torch.float16 and wt.dtype == torch.float16 and a_lo8.dtype == torch.uint8
    if w8 is None:
        raise ValueError("gemm16_lo8: the packed fp8 weight plane w8 = pack_weight_lo8(wt, sw) is required")
    m = a_hi.shape[0] if m is None else m
    kp = wt.shape[1]
    assert a_hi.shape[1] >= kp and a_lo8.shape[1] >= kp and wt.shape[0] >= n
    max_rows = lo8_max_rows(a_hi.stride(0), a_lo8.stride(0))
    if m > max_rows:
        for r0 in range(0, m, max_rows):
            r1 = min(m, r0 + max_rows)
            gemm16_lo8(a_hi[r0:r1], a_lo8[r0:r1], wt, sw, bias, n, epilogue, c=c[r0:r1] if c is not None else None,
                       resid=resid[r0:r1] if resid is not None else None, out_hi=out_hi[r0:r1] if out_hi is not None else None,
                       out_lo8=out_lo8[r0:r1] if out_lo8 is not None else None, m=r1 - r0, sa=sa, w8=w8)
        return
    with _timed("gemm_lo8_f16", 2.0 * m * n * kp):
        check(_lib.lib().llark_gemm16_lo8(
            epilogue, _dev(a_hi, "a_hi"), _dev(a_lo8, "a_lo8"), a_hi.stride(0), a_lo8.stride(0), _dev(wt, "wt"), wt.stride(0),
            _dev(w8, "w8", torch.uint8), w8.stride(0),
            _dev(bias, "bias", torch.float32) if bias is not None else None, m, n, kp, int(sa), int(sw),
            _dev(c, "c", torch.float32) if c is not None else None, c.stride(0) if c is not None else 0,
            _dev(resid, "resid", torch.float32) if resid is not None else None, resid.stride(0) if resid is not None else 0,
            _dev(out_hi, "out_hi", torch.float16) if out_hi is not None else None,
            _dev(out_lo8, "out_lo8", torch.uint8) if out_lo8 is not None else None,
            out_hi.stride(0) if out_hi is not None else 0, out_lo8.stride(0) if out_lo8 is not None else 0, workspace(), _stream()),
            "gemm16_lo8")


def lo8_max_rows(lda: int, lda8: int) -> int:
    """Largest row count one llark_gemm16_lo8 launch takes (32-bit byte offsets into the fp16 and e4m3 A planes), rounded down
    to whole 256-row tiles."""
    rows = min(((1 << 31) - 1) // (2 * lda), ((1 << 31) - 1) // max(1, lda8))
    return max(256, rows // 256 * 256)


def gemm16_resid_rmsnorm(a_hi: torch.Tensor, a_lo: Optional[torch.Tensor], wt: torch.Tensor, h: torch.Tensor, norm_w: torch.Tensor,
                         eps: float, x_hi: torch.Tensor, x_lo: Optional[torch.Tensor] = None) -> None:
    """Decode step: h += a . wt^T, then x = RMSNorm(h) as bf16 planes -- one launch (m <= 16)."""
    m, n, kp = a_hi.shape[0], h.shape[1], wt.shape[1]
    bf = torch.bfloat16
    name = ("gemm_split_" if a_lo is not None else "gemm_") + "bf16_skinny"
    with _timed(name, 2.0 * m * n * kp):
        check(_lib.lib().llark_gemm16_resid_rmsnorm(
            _DT[bf], int(a_lo is not None), _dev(a_hi, "a_hi", bf), _opt(a_lo, "a_lo", bf), a_hi.stride(0), _dev(wt, "wt", bf),
            wt.stride(0), m, n, kp, _dev(h, "h", torch.float32), h.stride(0), _dev(norm_w, "norm_w", torch.float32), float(eps),
            _dev(x_hi, "x_hi", bf), _opt(x_lo, "x_lo", bf), x_hi.stride(0), _stream()), "gemm16_resid_rmsnorm")


def gemv_dma_rmsnorm_takes(m: int, n: int, kp: int) -> bool:
    """RMSNorm + decode Linear in one launch of the streaming kernel (csrc/gemv_dma.hip): the row must live in registers."""
    return GEMV_DMA and m == 1 and kp <= 4096 and kp % 8 == 0 and n * kp * 2 >= GEMV_DMA_MIN_BYTES


def gemm16_rmsnorm_a(x: torch.Tensor, norm_w: torch.Tensor, eps: float, wt: torch.Tensor, n: int, epilogue: int, split: bool,
                     c: Optional[torch.Tensor] = None, out_hi: Optional[torch.Tensor] = None, out_lo: Optional[torch.Tensor] = None) -> None:
    """Decode step: RMSNorm(x) . wt^T in one launch (x fp32 [m <= 16][k]); epilogue EPI_F32 or SwiGLU."""
    m, kp = x.shape[0], wt.shape[1]
    bf = torch.bfloat16
    name = ("gemm_split_" if split else "gemm_") + "bf16_skinny"
    if gemv_dma_rmsnorm_takes(m, n, kp):
        with _timed(name, 2.0 * m * n * kp):
            check(_lib.lib().llark_gemv16_dma_rmsnorm(
                int(split), epilogue, _dev(x, "x", torch.float32), x.stride(0), _dev(norm_w, "norm_w", torch.float32), float(eps),
                _dev(wt, "wt", bf), wt.stride(0), None, m, n, kp, _opt(c, "c", torch.float32), c.stride(0) if c is not None else 0,
                _opt(out_hi, "out_hi", bf), _opt(out_lo, "out_lo", bf), out_hi.stride(0) if out_hi is not None else 0, _stream()),
                "gemv16_dma_rmsnorm")
        return
    with _timed(name, 2.0 * m * n * kp):
        check(_lib.lib().llark_gemm16_rmsnorm_a(
            _DT[bf], int(split), epilogue, _dev(x, "x", torch.float32), x.stride(0), _dev(norm_w, "norm_w", torch.float32), float(eps),
            _dev(wt, "wt", bf), wt.stride(0), None, m, n, kp, _opt(c, "c", torch.float32), c.stride(0) if c is not None else 0,
            _opt(out_hi, "out_hi", bf), _opt(out_lo, "out_lo", bf), out_hi.stride(0) if out_hi is not None else 0, _stream()),
            "gemm16_rmsnorm_a")


def pack_weight16_frag(wt: torch.Tensor, n: int) -> torch.Tensor:
    """wt [>=n][kp] 16-bit (pack_weight16 output, kp % 64 == 0) -> fragment-major copy for gemm16_fragw."""
    kp = wt.shape[1]
    out = torch.empty((round_up(n, 32) * kp,), dtype=wt.dtype, device=wt.device)
    check(_lib.lib().llark_pack_weight16_frag(_dev(wt, "wt"), wt.stride(0), n, kp, _dev(out, "out"), _stream()), "pack_weight16_frag")
    return out


def gemm16_fragw(a_hi: torch.Tensor, a_lo: Optional[torch.Tensor], wfrag: torch.Tensor, bias: Optional[torch.Tensor], n: int, kp: int,
                 epilogue: int, c: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
                 out_hi: Optional[torch.Tensor] = None, out_lo: Optional[torch.Tensor] = None, m: Optional[int] = None,
                 variant: int = -1, stream_k: Optional[bool] = None) -> None:
    """Same product / epilogues as gemm16, weights given fragment-major (pack_weight16_frag).
    variant: -1 library choice, 0 = 128x256 tiles, 1 = 128x128 tiles (no SwiGLU epilogue), 2 = 128x256 tiles with A by LDS-DMA
    (csrc/gemm_bda.hip: hi + lo bf16 operands only; bit-identical to 0).
    stream_k: None = library choice (split operands and less than one round of tiles: Llama o_proj / down_proj at M = 2968),
    True = cut whenever the tile count is not a whole number of rounds (forces the 128x256 tiles), False = off."""
    dtype = a_hi.dtype
    assert dtype in (torch.float16, torch.bfloat16) and wfrag.dtype == dtype and a_hi.shape[1] >= kp
    assert wfrag.numel() == round_up(n, 32) * kp
    m = a_hi.shape[0] if m is None else m
    name = ("gemm_split_" if a_lo is not None else "gemm_") + ("f16" if dtype == torch.float16 else "bf16")
    scratch = sk_scratch() if (stream_k is not False and variant <= 0) else None
    if stream_k and scratch is not None:
        variant = 0
    with _timed(name, 2.0 * m * n * kp):
      check(_lib.lib().llark_gemm16_fragw_sk(
        variant, _DT[dtype], int(a_lo is not None), epilogue, _dev(a_hi, "a_hi"), _dev(a_lo, "a_lo", dtype) if a_lo is not None else None,
        a_hi.stride(0), _dev(wfrag, "wfrag"), _dev(bias, "bias", torch.float32) if bias is not None else None,
        m, n, kp, _dev(c, "c", torch.float32) if c is not None else None, c.stride(0) if c is not None else 0,
        _dev(resid, "resid", torch.float32) if resid is not None else None, resid.stride(0) if resid is not None else 0,
        _dev(out_hi, "out_hi", dtype) if out_hi is not None else None,
        _dev(out_lo, "out_lo", dtype) if out_lo is not None else None,
        out_hi.stride(0) if out_hi is not None else 0,
        scratch.data_ptr() if scratch is not None else None, scratch.numel() * 4 if scratch is not None else 0, _stream()), "gemm16_fragw")


def gemm16_fragw_whole_tiles(split: bool, epilogue: int, m: int, n: int, kp: int) -> bool:
    """Does the library's own choice run this fragment-major product as whole 128x256 tiles (llark_gemm16_fragw_whole_tiles)?"""
    return bool(_lib.lib().llark_gemm16_fragw_whole_tiles(int(bool(split)), int(epilogue), int(m), int(n), int(kp)))


def rope_qkv_row_order(nh: int, hd: int = 128) -> torch.Tensor:
    """Row order of the fused q|k|v weight for llark_gemm16_fragw_rope_qkv: inside every q and k head the rows go
    [0..31 | 64..95 | 32..63 | 96..127] (a rotation pair d, d + 64 then sits in MFMA tiles 2j, 2j + 1 of one wave: same lane, same
    register); v rows keep their order.  Returns the int64 gather index over the 3 * nh * hd rows (a permutation)."""
    assert hd == 128
    inside = torch.cat((torch.arange(0, 32), torch.arange(64, 96), torch.arange(32, 64), torch.arange(96, 128)))
    qk = (torch.arange(2 * nh)[:, None] * hd + inside[None, :]).reshape(-1)
    return torch.cat((qk, torch.arange(2 * nh * hd, 3 * nh * hd)))


def gemm16_fragw_rope_qkv(a_hi: torch.Tensor, a_lo: Optional[torch.Tensor], wfrag: torch.Tensor, kp: int, batch: int, s: int, nh: int,
                          pos0: int, cos_t: torch.Tensor, sin_t: torch.Tensor, q: torch.Tensor, k_cache: torch.Tensor, vt_cache: torch.Tensor,
                          q_lo=None, k_cache_lo=None, vt_cache_lo=None) -> None:
    """Prefill q|k|v product with RoPE + head split + both cache writes in the epilogue (include/llark_hip.h,
    llark_gemm16_fragw_rope_qkv); ``wfrag`` = pack_weight16_frag of the weight gathered with :func:`rope_qkv_row_order`."""
    bf = torch.bfloat16
    hd = 128
    smax = k_cache.shape[-2]
    n = 3 * nh * hd
    assert a_hi.dtype == bf and wfrag.dtype == bf and wfrag.numel() == n * kp and a_hi.shape[0] >= batch * s and a_hi.shape[1] >= kp
    assert k_cache.shape[-1] == hd and vt_cache.shape[-1] == smax and q.numel() >= batch * nh * s * hd
    name = "gemm_split_bf16" if a_lo is not None else "gemm_bf16"
    with _timed(name, 2.0 * batch * s * n * kp):
        check(_lib.lib().llark_gemm16_fragw_rope_qkv(
            _dev(a_hi, "a_hi", bf), _opt(a_lo, "a_lo", bf), a_hi.stride(0), _dev(wfrag, "wfrag", bf), kp, batch, s, nh, hd, pos0,
            _dev(cos_t, "cos", torch.float32), _dev(sin_t, "sin", torch.float32), cos_t.shape[0], _dev(q, "q", bf),
            _dev(k_cache, "k_cache", bf), _dev(vt_cache, "vt_cache", bf), _opt(q_lo, "q_lo", bf), _opt(k_cache_lo, "k_cache_lo", bf),
            _opt(vt_cache_lo, "vt_cache_lo", bf), smax, _stream()), "gemm16_fragw_rope_qkv")


# ------------------------------------------------------------------------------------------------
# Llama
# ------------------------------------------------------------------------------------------------
def embed_gather(ids: torch.Tensor, table: torch.Tensor, out: torch.Tensor) -> None:
    """out[row] = table[ids[row]] ; ids flat int64 [rows]; out fp32 [rows][>=width]."""
    rows = ids.numel()
    vocab, width = table.shape
    assert out.shape[0] == rows and out.shape[1] >= width
    check(_lib.lib().llark_embed_gather(_dev(ids, "ids", torch.int64), rows, _dev(table, "table"), _DT[table.dtype], vocab,
                                        width, _dev(out, "out", torch.float32), out.stride(0), _stream()), "embed_gather")


def rmsnorm_bf16(x: torch.Tensor, w: torch.Tensor, eps: float, out_hi: torch.Tensor, out_lo: Optional[torch.Tensor] = None):
    rows, width = x.shape
    check(_lib.lib().llark_rmsnorm_bf16(_dev(x, "x", torch.float32), x.stride(0), rows, width, _dev(w, "w", torch.float32),
                                        float(eps), _dev(out_hi, "out_hi", torch.bfloat16),
                                        _dev(out_lo, "out_lo", torch.bfloat16) if out_lo is not None else None,
                                        out_hi.stride(0), _stream()), "rmsnorm_bf16")


def _opt(t, name, dtype):
    return _dev(t, name, dtype) if t is not None else None


def rope_split_heads(qkv: torch.Tensor, batch: int, s: int, nh: int, hd: int, pos0: int, cos_t: torch.Tensor,
                     sin_t: torch.Tensor, q: torch.Tensor, k_cache: torch.Tensor, vt_cache: torch.Tensor,
                     q_lo=None, k_cache_lo=None, vt_cache_lo=None) -> None:
    smax = k_cache.shape[-2]
    assert qkv.shape == (batch * s, 3 * nh * hd) and k_cache.shape[-1] == hd and vt_cache.shape[-1] == smax
    bf = torch.bfloat16
    check(_lib.lib().llark_rope_split_heads(_dev(qkv, "qkv", torch.float32), batch, s, nh, hd, pos0,
                                            _dev(cos_t, "cos", torch.float32), _dev(sin_t, "sin", torch.float32),
                                            cos_t.shape[0], _dev(q, "q", bf), _dev(k_cache, "k_cache", bf),
                                            _dev(vt_cache, "vt_cache", bf), _opt(q_lo, "q_lo", bf),
                                            _opt(k_cache_lo, "k_cache_lo", bf), _opt(vt_cache_lo, "vt_cache_lo", bf), smax,
                                            _stream()), "rope_split_heads")


def rope_split_heads_dpos(qkv, batch: int, nh: int, hd: int, pos_dev: torch.Tensor, cos_t, sin_t, q, k_cache, vt_cache,
                          q_lo=None, k_cache_lo=None, vt_cache_lo=None) -> None:
    """Decode step (s = 1) with the position in device memory (int32 scalar tensor): graph-capturable."""
    smax = k_cache.shape[-2]
    bf = torch.bfloat16
    check(_lib.lib().llark_rope_split_heads_dpos(_dev(qkv, "qkv", torch.float32), batch, nh, hd, _dev(pos_dev, "pos", torch.int32),
                                                 _dev(cos_t, "cos", torch.float32), _dev(sin_t, "sin", torch.float32),
                                                 _dev(q, "q", bf), _dev(k_cache, "k_cache", bf), _dev(vt_cache, "vt_cache", bf),
                                                 _opt(q_lo, "q_lo", bf), _opt(k_cache_lo, "k_cache_lo", bf),
                                                 _opt(vt_cache_lo, "vt_cache_lo", bf), smax, _stream()), "rope_split_heads_dpos")


def attn_decode_dpos(q, k_cache, vt_cache, batch: int, nh: int, hd: int, pos_dev: torch.Tensor, out,
                     q_lo=None, k_cache_lo=None, vt_cache_lo=None, out_lo=None) -> None:
    smax = k_cache.shape[-2]
    bf = torch.bfloat16
    check(_lib.lib().llark_attn_decode_bf16_dpos(_dev(q, "q", bf), _dev(k_cache, "k_cache", bf), _dev(vt_cache, "vt_cache", bf),
                                                 _opt(q_lo, "q_lo", bf), _opt(k_cache_lo, "k_cache_lo", bf),
                                                 _opt(vt_cache_lo, "vt_cache_lo", bf), batch, nh, hd,
                                                 _dev(pos_dev, "pos", torch.int32), smax, _dev(out, "out", bf),
                                                 _opt(out_lo, "out_lo", bf), _stream()), "attn_decode_dpos")


def attn_prefill(q, k_cache, vt_cache, batch: int, s: int, nh: int, hd: int, past: int, out: torch.Tensor,
                 q_lo=None, k_cache_lo=None, vt_cache_lo=None, out_lo=None, alibi_slopes=None) -> None:
    smax = k_cache.shape[-2]
    bf = torch.bfloat16
    check(_lib.lib().llark_attn_prefill_bf16_alibi(_dev(q, "q", bf), _dev(k_cache, "k_cache", bf), _dev(vt_cache, "vt_cache", bf),
                                                   _opt(q_lo, "q_lo", bf), _opt(k_cache_lo, "k_cache_lo", bf),
                                                   _opt(vt_cache_lo, "vt_cache_lo", bf), batch, s, nh, hd, past, smax,
                                                   _dev(out, "out", bf), _opt(out_lo, "out_lo", bf),
                                                   _opt(alibi_slopes, "alibi_slopes", torch.float32), _stream()), "attn_prefill")


def attn_prefill_lse(q, k_cache, vt_cache, batch: int, s: int, nh: int, hd: int, out: torch.Tensor, lse: torch.Tensor,
                     alibi_slopes=None) -> None:
    """Training forward: causal attention without past keys that also leaves each query's log-sum-exp (fp32 [batch*nh][s])."""
    smax = k_cache.shape[-2]
    bf = torch.bfloat16
    check(_lib.lib().llark_attn_prefill_bf16_lse(_dev(q, "q", bf), _dev(k_cache, "k_cache", bf), _dev(vt_cache, "vt_cache", bf),
                                                 batch, s, nh, hd, smax, _dev(out, "out", bf), _dev(lse, "lse", torch.float32),
                                                 _opt(alibi_slopes, "alibi_slopes", torch.float32), _stream()), "attn_prefill_lse")


def attn_backward(q, k_cache, v_rm, dO, o, lse, dsum, batch: int, s: int, nh: int, hd: int,
                  dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, alibi_slopes=None) -> None:
    """Flash-style backward of causal attention (csrc/attn_bwd.hip); layouts in include/llark_hip.h."""
    smax = k_cache.shape[-2]
    bf, f32 = torch.bfloat16, torch.float32
    check(_lib.lib().llark_attn_backward_bf16(_dev(q, "q", bf), _dev(k_cache, "k_cache", bf), _dev(v_rm, "v_rm", bf), _dev(dO, "dO", bf),
                                              _dev(o, "o", bf), _dev(lse, "lse", f32), _dev(dsum, "dsum", f32), batch, s, nh, hd, smax,
                                              _dev(dq, "dq", f32), _dev(dk, "dk", f32), _dev(dv, "dv", f32),
                                              _opt(alibi_slopes, "alibi_slopes", f32), _stream()), "attn_backward")


def attn_decode(q, k_cache, vt_cache, batch: int, nh: int, hd: int, total: int, out: torch.Tensor,
                q_lo=None, k_cache_lo=None, vt_cache_lo=None, out_lo=None, alibi_slopes=None) -> None:
    smax = k_cache.shape[-2]
    bf = torch.bfloat16
    check(_lib.lib().llark_attn_decode_bf16_alibi(_dev(q, "q", bf), _dev(k_cache, "k_cache", bf), _dev(vt_cache, "vt_cache", bf),
                                                  _opt(q_lo, "q_lo", bf), _opt(k_cache_lo, "k_cache_lo", bf),
                                                  _opt(vt_cache_lo, "vt_cache_lo", bf), batch, nh, hd, total, smax,
                                                  _dev(out, "out", bf), _opt(out_lo, "out_lo", bf),
                                                  _opt(alibi_slopes, "alibi_slopes", torch.float32), _stream()), "attn_decode")


def attn_decode_rope(qkv: torch.Tensor, batch: int, nh: int, hd: int, pos, cos_t, sin_t, k_cache, vt_cache, out,
                     k_cache_lo=None, vt_cache_lo=None, out_lo=None, alibi_slopes=None) -> None:
    """Decode step in one launch: RoPE of the new token + KV-cache append + attention.  `pos` is an int or an int32 scalar
    device tensor (position read on the device: replayable launch lists / captured graphs)."""
    smax = k_cache.shape[-2]
    bf = torch.bfloat16
    dpos = isinstance(pos, torch.Tensor)
    assert qkv.shape == (batch, 3 * nh * hd)
    check(_lib.lib().llark_attn_decode_rope_bf16(_dev(qkv, "qkv", torch.float32), batch, nh, hd, 0 if dpos else int(pos),
                                                 _dev(pos, "pos", torch.int32) if dpos else None, _dev(cos_t, "cos", torch.float32),
                                                 _dev(sin_t, "sin", torch.float32), cos_t.shape[0], _dev(k_cache, "k_cache", bf),
                                                 _dev(vt_cache, "vt_cache", bf), _opt(k_cache_lo, "k_cache_lo", bf),
                                                 _opt(vt_cache_lo, "vt_cache_lo", bf), smax, _dev(out, "out", bf), _opt(out_lo, "out_lo", bf),
                                                 _opt(alibi_slopes, "alibi_slopes", torch.float32), _stream()), "attn_decode_rope")


# ------------------------------------------------------------------------------------------------
# MPT
# ------------------------------------------------------------------------------------------------
def layernorm_bf16(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], eps: float, out_hi: torch.Tensor,
                   out_lo: Optional[torch.Tensor] = None) -> None:
    rows, width = x.shape
    bf = torch.bfloat16
    check(_lib.lib().llark_layernorm_bf16(_dev(x, "x", torch.float32), x.stride(0), rows, width, _dev(gamma, "gamma", torch.float32),
                                          _opt(beta, "beta", torch.float32), float(eps), _dev(out_hi, "out_hi", bf),
                                          _opt(out_lo, "out_lo", bf), out_hi.stride(0), _stream()), "layernorm_bf16")


def layernorm_f32_(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], eps: float) -> None:
    """In-place LayerNorm of a (possibly column-sliced) fp32 matrix: rows of x.shape[1] values with stride x.stride(0)."""
    if x.stride(1) != 1:
        raise ValueError("layernorm_f32_: rows must be contiguous")
    if not x.is_cuda:
        raise _lib.LlarkHipError("x: tensor must live on the GPU (there is no CPU fallback)")
    rows, width = x.shape
    check(_lib.lib().llark_layernorm_f32(x.data_ptr(), x.stride(0), rows, width, _dev(gamma, "gamma", torch.float32),
                                         _opt(beta, "beta", torch.float32), float(eps), x.data_ptr(), x.stride(0), _stream()),
          "layernorm_f32")


def clamp_f32_(x: torch.Tensor, limit: float) -> None:
    check(_lib.lib().llark_clamp_f32(_dev(x, "x", torch.float32), x.numel(), float(limit), _stream()), "clamp_f32")


def clamp_bwd_bf16_(x_pre: torch.Tensor, limit: float, dy: torch.Tensor) -> None:
    """dy (bf16, in place) <- dy where |x_pre| <= limit else 0: the gradient of clamp_f32_ (x_pre = the values before it)."""
    assert x_pre.numel() == dy.numel()
    check(_lib.lib().llark_clamp_bwd_bf16(_dev(x_pre, "x_pre", torch.float32), x_pre.numel(), float(limit),
                                          _dev(dy, "dy", torch.bfloat16), _stream()), "clamp_bwd_bf16")


def layernorm_bwd(x: torch.Tensor, gamma: torch.Tensor, dy: torch.Tensor, eps: float, dx: torch.Tensor, dgamma: torch.Tensor,
                  dbeta: Optional[torch.Tensor], accumulate: bool) -> None:
    """x, dy, dx: fp32 [rows][width] views with unit column stride (column slices of wider buffers are fine)."""
    rows, width = x.shape
    for t, nm in ((x, "x"), (dy, "dy"), (dx, "dx")):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(1) != 1:
            raise _lib.LlarkHipError(f"layernorm_bwd: {nm} must be an fp32 GPU tensor with contiguous rows")
    check(_lib.lib().llark_layernorm_bwd(x.data_ptr(), x.stride(0), _dev(gamma, "gamma", torch.float32), dy.data_ptr(), dy.stride(0), rows, width,
                                         float(eps), dx.data_ptr(), dx.stride(0), _dev(dgamma, "dgamma", torch.float32),
                                         _opt(dbeta, "dbeta", torch.float32), int(accumulate), _stream()), "layernorm_bwd")


def gelu_bwd(up: torch.Tensor, dact: torch.Tensor, dup16: torch.Tensor, dup32: Optional[torch.Tensor] = None) -> None:
    check(_lib.lib().llark_gelu_bwd(_dev(up, "up", torch.float32), _dev(dact, "dact", torch.float32), up.numel(),
                                    _opt(dup32, "dup32", torch.float32), _dev(dup16, "dup16", torch.bfloat16), _stream()), "gelu_bwd")


def causal_softmax_rows_alibi(scores: torch.Tensor, batch: int, s: int, scale: float, slopes: torch.Tensor, nh: int, p_out: torch.Tensor) -> None:
    check(_lib.lib().llark_causal_softmax_rows_alibi(_dev(scores, "scores", torch.float32), batch, s, float(scale),
                                                     _dev(slopes, "slopes", torch.float32), nh, _dev(p_out, "p", torch.bfloat16),
                                                     p_out.stride(-2), _stream()), "causal_softmax_rows_alibi")


def scale_f32_(x: torch.Tensor, a: float) -> None:
    check(_lib.lib().llark_scale_f32(_dev(x, "x", torch.float32), x.numel(), float(a), _stream()), "scale_f32")


def gelu_split_bf16(x: torch.Tensor, out_hi: torch.Tensor, out_lo: Optional[torch.Tensor] = None) -> None:
    rows, width = x.shape
    bf = torch.bfloat16
    check(_lib.lib().llark_gelu_split_bf16(_dev(x, "x", torch.float32), x.stride(0), rows, width, _plane(out_hi, "out_hi"),
                                           _plane(out_lo, "out_lo", out_hi), out_hi.stride(0), _stream()), "gelu_split_bf16")


def cross_entropy_shifted(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """Mean NLL of logits[:, :-1] against labels[:, 1:] ignoring ``ignore_index`` (scalar device tensor)."""
    B, S, V = logits.shape
    assert labels.shape == (B, S)
    row_loss = torch.empty((B * S,), dtype=torch.float32, device=logits.device)
    out = torch.empty((2,), dtype=torch.float32, device=logits.device)
    check(_lib.lib().llark_cross_entropy_shifted(_dev(logits, "logits", torch.float32), logits.stride(1), B, S, V,
                                                 _dev(labels.contiguous(), "labels", torch.int64), ignore_index,
                                                 _dev(row_loss, "row_loss"), _dev(out, "loss"), _stream()),
          "cross_entropy_shifted")
    return out[0]


# ------------------------------------------------------------------------------------------------
# training step pieces
# ------------------------------------------------------------------------------------------------
def gemm16_batched(a: torch.Tensor, stride_a: int, lda: int, wt: torch.Tensor, stride_w: int, ldw: int, m: int, n: int,
                   kp: int, batch: int, c: torch.Tensor, ldc: int, stride_c: int) -> None:
    """fp32 C[b] = A[b] . Wt[b]^T for b < batch (bf16 operands given as flat tensors + element strides)."""
    check(_lib.lib().llark_gemm16_batched(_DT[a.dtype], 0, EPI_F32, _dev(a, "a"), None, lda, stride_a, _dev(wt, "wt"), ldw,
                                          stride_w, m, n, kp, _dev(c, "c", torch.float32), ldc, stride_c, None, None, 0, 0,
                                          batch, _stream()), "gemm16_batched")


def transpose16(src: torch.Tensor, ld_src: int, rows: int, cols: int, dst: torch.Tensor, ld_dst: int, batch: int = 1,
                stride_src: int = 0, stride_dst: int = 0) -> None:
    check(_lib.lib().llark_transpose16(_dev(src, "src"), ld_src, rows, cols, _dev(dst, "dst"), ld_dst, batch, stride_src,
                                       stride_dst, _stream()), "transpose16")


def transposed16(x: torch.Tensor, kmult: int = 64) -> torch.Tensor:
    """[R][C] 16-bit -> new [C][round_up(R)] (zero padded): the K-contiguous operand of a dW / dX GEMM."""
    R, C = x.shape
    out = torch.empty((C, round_up(R, kmult)), dtype=x.dtype, device=x.device)
    transpose16(x, x.stride(0), R, C, out, out.shape[1])
    return out


def split_heads16(x: torch.Tensor, batch: int, s: int, nh: int, hd: int, out: torch.Tensor) -> None:
    check(_lib.lib().llark_split_heads16(_dev(x, "x"), batch, s, nh, hd, _dev(out, "out"), _stream()), "split_heads16")


def causal_softmax_rows(scores: torch.Tensor, batch: int, s: int, scale: float, p_out: torch.Tensor) -> None:
    check(_lib.lib().llark_causal_softmax_rows(_dev(scores, "scores", torch.float32), batch, s, float(scale),
                                               _dev(p_out, "p", torch.bfloat16), p_out.shape[-1], _stream()), "causal_softmax_rows")


def attn_ds(p: torch.Tensor, dp: torch.Tensor, batch: int, s: int, scale: float, ds_out: torch.Tensor) -> None:
    check(_lib.lib().llark_attn_ds(_dev(p, "p", torch.bfloat16), _dev(dp, "dp", torch.float32), batch, s, float(scale),
                                   _dev(ds_out, "ds", torch.bfloat16), p.shape[-1], _stream()), "attn_ds")


def rope_merge_bwd(dq, dk, dv, cos_t, sin_t, batch: int, s: int, nh: int, hd: int, pos0: int, dqkv: torch.Tensor) -> None:
    f = torch.float32
    check(_lib.lib().llark_rope_merge_bwd(_dev(dq, "dq", f), _dev(dk, "dk", f), _dev(dv, "dv", f), _dev(cos_t, "cos", f),
                                          _dev(sin_t, "sin", f), batch, s, nh, hd, pos0, _dev(dqkv, "dqkv", torch.bfloat16),
                                          _stream()), "rope_merge_bwd")


def rmsnorm_bwd(x, w, dy, eps: float, dx, accumulate: bool, dw, dx16: Optional[torch.Tensor] = None) -> None:
    """RMSNorm backward (include/llark_hip.h: llark_rmsnorm_bwd); ``dx16`` (bf16 [rows][>= width]) also receives the final dx rounded
    to bf16 -- the next products' A operand, without a split16 pass (llark_rmsnorm_bwd_out16)."""
    rows, width = x.shape
    f = torch.float32
    if dx16 is not None:
        assert dx16.dtype == torch.bfloat16 and dx16.shape[0] >= rows and dx16.stride(1) == 1
        check(_lib.lib().llark_rmsnorm_bwd_out16(_dev(x, "x", f), _dev(w, "w", f), _dev(dy, "dy", f), rows, width, float(eps),
                                                 _dev(dx, "dx", f), int(accumulate), _dev(dw, "dw", f),
                                                 _dev(dx16, "dx16", torch.bfloat16, contiguous=False), dx16.stride(0), _stream()), "rmsnorm_bwd_out16")
        return
    check(_lib.lib().llark_rmsnorm_bwd(_dev(x, "x", f), _dev(w, "w", f), _dev(dy, "dy", f), rows, width, float(eps),
                                       _dev(dx, "dx", f), int(accumulate), _dev(dw, "dw", f), _stream()), "rmsnorm_bwd")


def swiglu_fwd(gu: torch.Tensor, act: torch.Tensor) -> None:
    rows, two_i = gu.shape
    check(_lib.lib().llark_swiglu_fwd(_dev(gu, "gu", torch.float32), rows, two_i // 2, _dev(act, "act", torch.bfloat16),
                                      _stream()), "swiglu_fwd")


def swiglu_bwd(gu: torch.Tensor, dact: torch.Tensor, dgu: torch.Tensor) -> None:
    rows, two_i = gu.shape
    check(_lib.lib().llark_swiglu_bwd(_dev(gu, "gu", torch.float32), _dev(dact, "dact", torch.float32), rows, two_i // 2,
                                      _dev(dgu, "dgu", torch.bfloat16), _stream()), "swiglu_bwd")


def cross_entropy_fwd_bwd(logits: torch.Tensor, labels: torch.Tensor, dlogits: torch.Tensor, loss_scale: float = 1.0,
                          ignore_index: int = -100) -> torch.Tensor:
    """Mean shifted CE (device scalar) and its gradient w.r.t. the logits as bf16 [B*S][ldd]."""
    B, S, V = logits.shape
    row_loss = torch.empty((B * S,), dtype=torch.float32, device=logits.device)
    out = torch.empty((2,), dtype=torch.float32, device=logits.device)
    lab = labels.contiguous()
    check(_lib.lib().llark_cross_entropy_shifted(_dev(logits, "logits", torch.float32), logits.stride(1), B, S, V,
                                                 _dev(lab, "labels", torch.int64), ignore_index, _dev(row_loss, "row_loss"),
                                                 _dev(out, "loss"), _stream()), "cross_entropy_shifted")
    check(_lib.lib().llark_cross_entropy_bwd(_dev(logits, "logits"), logits.stride(1), B, S, V, _dev(lab, "labels"),
                                             _dev(row_loss, "row_loss"), _dev(out, "loss"), float(loss_scale),
                                             _dev(dlogits, "dlogits", torch.bfloat16), dlogits.stride(0), _stream()),
          "cross_entropy_bwd")
    return out[0]


def sumsq_f32(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """sum of squares of a contiguous fp32 tensor as a device double scalar (the squared gradient norm of clip_grad_norm_);
    ``out`` + ``accumulate``: add to an existing scalar (per-slice partial norms)."""
    if out is None:
        out = torch.empty((1,), dtype=torch.float64, device=x.device)
        accumulate = False
    check(_lib.lib().llark_sumsq_f32(_dev(x, "x", torch.float32), x.numel(), _dev(out, "out", torch.float64), int(accumulate), _stream()),
          "sumsq_f32")
    return out


def colsum_add(x: torch.Tensor, out: torch.Tensor) -> None:
    rows, cols = x.shape
    check(_lib.lib().llark_colsum_f32(_dev(x, "x", torch.float32), x.stride(0), rows, cols, _dev(out, "out", torch.float32),
                                      _stream()), "colsum")


def gather_rows(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor) -> None:
    check(_lib.lib().llark_gather_rows_f32(_dev(src, "src", torch.float32), src.stride(0), _dev(idx, "idx", torch.int64),
                                           idx.numel(), src.shape[1], _dev(dst, "dst", torch.float32), dst.stride(0), _stream()),
          "gather_rows")


def scatter_add_rows(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor) -> None:
    check(_lib.lib().llark_scatter_add_rows_f32(_dev(src, "src", torch.float32), src.stride(0), _dev(idx, "idx", torch.int64),
                                                idx.numel(), src.shape[1], _dev(dst, "dst", torch.float32), dst.stride(0),
                                                _stream()), "scatter_add_rows")


def adamw(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
          weight_decay: float, step: int, grad_scale: float = 1.0, grad_sumsq: Optional[torch.Tensor] = None,
          max_grad_norm: float = 0.0) -> None:
    """torch.optim.AdamW step; with ``grad_sumsq`` (device double from sumsq_f32 over the WHOLE gradient) and ``max_grad_norm`` the
    clip_grad_norm_ coefficient is computed and applied on the device (include/llark_hip.h: llark_adamw_clip)."""
    assert p.numel() == g.numel() == m.numel() == v.numel()
    args = (_DT[p.dtype], _dev(p, "p"), _dev(g, "g", torch.float32), _dev(m, "m", torch.float32), _dev(v, "v", torch.float32), p.numel(),
            float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale))
    if grad_sumsq is None:
        check(_lib.lib().llark_adamw(*args, _stream()), "adamw")
    else:
        check(_lib.lib().llark_adamw_clip(*args, _dev(grad_sumsq, "grad_sumsq", torch.float64), float(max_grad_norm), _stream()), "adamw_clip")


def device_info(device: int = 0) -> Tuple[int, str]:
    import ctypes
    buf = ctypes.create_string_buffer(64)
    cus = _lib.lib().llark_device_info(device, buf, 64)
    if cus < 0:
        check(cus, "device_info")
    return cus, buf.value.decode()


# ---- CLAP HTSAT audio encoder (csrc/clap.hip) ----------------------------------------------------------------------------

def clap_patchify(x: torch.Tensor, bn_mean: torch.Tensor, bn_scale: torch.Tensor, bn_bias: torch.Tensor, tap_idx: torch.Tensor,
                  tap_w: torch.Tensor, spec: int, patch: int, out_hi: torch.Tensor, out_lo: Optional[torch.Tensor]) -> None:
    """log-mel x (B, frames, mel) -> BatchNorm -> bicubic time stretch -> fold -> patch rows (bf16 planes)."""
    B, frames, mel = x.shape
    assert tap_idx.shape == (spec * spec // mel, 4) and tap_w.shape == tap_idx.shape and out_hi.shape[0] == B * (spec // patch) ** 2
    bf = torch.bfloat16
    check(_lib.lib().llark_clap_patchify(_dev(x, "x", torch.float32), B, frames, mel, _dev(bn_mean, "bn_mean", torch.float32),
                                         _dev(bn_scale, "bn_scale", torch.float32), _dev(bn_bias, "bn_bias", torch.float32),
                                         _dev(tap_idx, "tap_idx", torch.int32), _dev(tap_w, "tap_w", torch.float32), spec, patch,
                                         _dev(out_hi, "out_hi", bf), _opt(out_lo, "out_lo", bf), out_hi.stride(0), _stream()),
          "clap_patchify")


def _plane(t: Optional[torch.Tensor], name: str, like: Optional[torch.Tensor] = None):
    """Pointer of a 2-D bf16 plane that may be a column block of a wider buffer (unit column stride, any row stride)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.LlarkHipError(f"{name}: tensor must live on the GPU (there is no CPU fallback)")
    if t.dtype != torch.bfloat16 or t.dim() != 2 or t.stride(1) != 1:
        raise TypeError(f"{name}: expected a 2-D bf16 plane with unit column stride")
    if like is not None and (t.stride(0) != like.stride(0) or t.shape != like.shape):
        raise ValueError(f"{name}: planes of one operand must share shape and row stride")
    return t.data_ptr()


def clap_window_attn(qkv: torch.Tensor, batch: int, H: int, W: int, C: int, heads: int, window: int, shift: int,
                     bias_table: torch.Tensor, out_hi: torch.Tensor, out_lo: Optional[torch.Tensor],
                     out_hi_dup: Optional[torch.Tensor] = None) -> None:
    """Swin window attention on a (batch, H, W) token map; qkv fp32 [batch*H*W][3C] in token order."""
    assert qkv.shape == (batch * H * W, 3 * C) and bias_table.shape == ((2 * window - 1) ** 2, heads) and out_hi.shape == (batch * H * W, C)
    with _timed("clap_window_attn", 4.0 * batch * H * W * window * window * C):
        check(_lib.lib().llark_clap_window_attn(_dev(qkv, "qkv", torch.float32), qkv.stride(0), batch, H, W, C, heads, window, shift,
                                                _dev(bias_table, "bias_table", torch.float32), _plane(out_hi, "out_hi"),
                                                _plane(out_lo, "out_lo", out_hi), _plane(out_hi_dup, "out_hi_dup", out_hi),
                                                out_hi.stride(0), _stream()), "clap_window_attn")


def layernorm_bf16_dup(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], eps: float, out_hi: torch.Tensor,
                       out_lo: torch.Tensor, out_hi_dup: torch.Tensor) -> None:
    """LayerNorm -> the three column blocks [hi | lo | hi] of a K-concatenated operand (see gemm16_act)."""
    rows, width = x.shape
    check(_lib.lib().llark_layernorm_bf16_dup(_dev(x, "x", torch.float32), x.stride(0), rows, width, _dev(gamma, "gamma", torch.float32),
                                              _opt(beta, "beta", torch.float32), float(eps), _plane(out_hi, "out_hi"),
                                              _plane(out_lo, "out_lo", out_hi), _plane(out_hi_dup, "out_hi_dup", out_hi),
                                              out_hi.stride(0), _stream()), "layernorm_bf16_dup")


def gemm16_act(a_hi: torch.Tensor, a_lo: Optional[torch.Tensor], wt: torch.Tensor, bias: Optional[torch.Tensor], n: int,
               out_hi: torch.Tensor, out_lo: Optional[torch.Tensor] = None, out_hi_dup: Optional[torch.Tensor] = None,
               act: int = 0, variant: int = -1) -> None:
    """(a_hi [+ a_lo]) . wt^T + bias -> optional exact GELU (act=2) -> bf16 planes (hi, or hi + lo [+ a second hi])."""
    bf = torch.bfloat16
    m, kp = a_hi.shape[0], wt.shape[1]
    assert a_hi.shape[1] >= kp and wt.shape[0] >= n and out_hi.shape == (m, n)
    epi = EPI_SPLIT16 if out_lo is not None else EPI_OUT16
    with _timed("gemm_split_bf16" if a_lo is not None else "gemm_bf16", 2.0 * m * n * kp):
        check(_lib.lib().llark_gemm16_act(variant, BF16, int(a_lo is not None), epi, _dev(a_hi, "a_hi", bf), _opt(a_lo, "a_lo", bf),
                                          a_hi.stride(0), _dev(wt, "wt", bf), wt.stride(0), _opt(bias, "bias", torch.float32), m, n, kp,
                                          _plane(out_hi, "out_hi"), _plane(out_lo, "out_lo", out_hi),
                                          _plane(out_hi_dup, "out_hi_dup", out_hi), out_hi.stride(0), act, _stream()), "gemm16_act")


def clap_patch_merge(x: torch.Tensor, batch: int, H: int, W: int, out: torch.Tensor) -> None:
    C = x.shape[1]
    assert x.shape[0] == batch * H * W and out.shape == (batch * H * W // 4, 4 * C)
    check(_lib.lib().llark_clap_patch_merge(_dev(x, "x", torch.float32), x.stride(0), batch, H, W, C, _dev(out, "out", torch.float32),
                                            out.stride(0), _stream()), "clap_patch_merge")


def mean_rows_f32(x: torch.Tensor, batch: int, out: torch.Tensor) -> None:
    rows, C = x.shape
    assert rows % batch == 0 and out.shape == (batch, C)
    check(_lib.lib().llark_mean_rows_f32(_dev(x, "x", torch.float32), x.stride(0), batch, rows // batch, C,
                                         _dev(out, "out", torch.float32), out.stride(0), _stream()), "mean_rows_f32")


def relu_split_bf16(x: torch.Tensor, out_hi: torch.Tensor, out_lo: Optional[torch.Tensor] = None) -> None:
    rows, width = x.shape
    bf = torch.bfloat16
    check(_lib.lib().llark_relu_split_bf16(_dev(x, "x", torch.float32), x.stride(0), rows, width, _dev(out_hi, "out_hi", bf),
                                           _opt(out_lo, "out_lo", bf), out_hi.stride(0), _stream()), "relu_split_bf16")


def l2_normalize_rows_(x: torch.Tensor, eps: float = 1e-12) -> None:
    rows, width = x.shape
    check(_lib.lib().llark_l2_normalize_rows(_dev(x, "x", torch.float32), x.stride(0), rows, width, float(eps), _stream()),
          "l2_normalize_rows")


def clap_logmel(wav: torch.Tensor, window: torch.Tensor, twiddle: torch.Tensor, melw: torch.Tensor, mel_lo: torch.Tensor,
                mel_hi: torch.Tensor, quantize_int16: bool = False) -> torch.Tensor:
    """wav (B, n) fp32 48 kHz -> (B, n // 480 + 1, 64) log-mel dB (fused STFT + mel + log kernel)."""
    B, n = wav.shape
    assert window.shape == (1024,) and twiddle.shape == (512, 2) and melw.shape == (64, 513) and mel_lo.shape == (64,) and mel_hi.shape == (64,)
    out = torch.empty((B, n // 480 + 1, 64), dtype=torch.float32, device=wav.device)
    with _timed("clap_logmel", 0.0):
        check(_lib.lib().llark_clap_logmel(_dev(wav, "wav", torch.float32), B, n, int(quantize_int16), _dev(window, "window", torch.float32),
                                           _dev(twiddle, "twiddle", torch.float32), _dev(melw, "melw", torch.float32),
                                           _dev(mel_lo, "mel_lo", torch.int32), _dev(mel_hi, "mel_hi", torch.int32),
                                           _dev(out, "out", torch.float32), _stream()), "clap_logmel")
    return out


def split16_into(x: torch.Tensor, hi: torch.Tensor, lo: Optional[torch.Tensor]) -> None:
    """split16 into caller-owned planes (row stride hi.stride(0) >= width; pad columns are the caller's to zero)."""
    rows, width = x.shape
    check(_lib.lib().llark_split16(_DT[hi.dtype], _dev(x, "x", torch.float32), x.stride(0), rows, width, _dev(hi, "hi"),
                                   _opt(lo, "lo", hi.dtype), hi.stride(0), _stream()), "split16")


def adamw_twins(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
                weight_decay: float, step: int, grad_scale: float = 1.0, grad_sumsq: Optional[torch.Tensor] = None,
                max_grad_norm: float = 0.0, wfrag: Optional[torch.Tensor] = None, rope_rows: int = 0,
                wtfrag: Optional[torch.Tensor] = None) -> None:
    """:func:`adamw` on a bf16 weight matrix ``p`` [n][k] that also writes the fragment-major twins of the updated weight
    (include/llark_hip.h: llark_adamw_twins): ``wfrag`` = pack_weight16_frag(p) (rows < ``rope_rows`` in :func:`rope_qkv_row_order`),
    ``wtfrag`` = pack_weight16_frag(p^T)."""
    n, k = p.shape
    bf = torch.bfloat16
    assert p.dtype == bf and p.is_contiguous() and g.numel() == m.numel() == v.numel() == n * k
    for t in (wfrag, wtfrag):
        assert t is None or (t.dtype == bf and t.numel() == n * k)
    check(_lib.lib().llark_adamw_twins(
        _dev(p, "p"), _dev(g, "g", torch.float32), _dev(m, "m", torch.float32), _dev(v, "v", torch.float32), n, k, float(lr), float(beta1),
        float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale),
        _dev(grad_sumsq, "grad_sumsq", torch.float64) if grad_sumsq is not None else None, float(max_grad_norm),
        _opt(wfrag, "wfrag", bf), int(rope_rows), _opt(wtfrag, "wtfrag", bf), _stream()), "adamw_twins")


def adamw_twins_takes(n: int, k: int) -> bool:
    """Shapes llark_adamw_twins takes with both twins."""
    return n % 64 == 0 and k % 128 == 0 and n // 32 <= 65535


def gemm16_fragw_swiglu_train(mode: int, a: torch.Tensor, wfrag: torch.Tensor, n: int, kp: int, out: torch.Tensor, gu16: torch.Tensor) -> bool:
    """The SwiGLU products of the training step with the element-wise pass in the epilogue (include/llark_hip.h:
    llark_gemm16_fragw_swiglu_train).  mode 0: out = act [m][I], gu16 [m][2 I] written; mode 1: out = d(gate | up) [m][2 I], gu16 read.
    Returns False (nothing launched) when the kernel does not take the shape: the caller keeps the two-launch path."""
    bf = torch.bfloat16
    m = a.shape[0]
    assert a.dtype == bf and wfrag.dtype == bf and out.dtype == bf and gu16.dtype == bf and a.shape[1] >= kp
    assert wfrag.numel() == round_up(n, 32) * kp
    if kp < 192 or m * a.stride(0) * 2 >= (1 << 31):
        return False
    name = "gemm_bf16"
    with _timed(name, 2.0 * m * n * kp):
        check(_lib.lib().llark_gemm16_fragw_swiglu_train(int(mode), _dev(a, "a", contiguous=False), a.stride(0), _dev(wfrag, "wfrag"), m, n, kp,
                                                         _dev(out, "out", contiguous=False), out.stride(0),
                                                         _dev(gu16, "gu16", contiguous=False), gu16.stride(0), _stream()), "gemm16_fragw_swiglu_train")
    return True


def pack_frag_t16(x: torch.Tensor, n: Optional[int] = None, chunk16: bool = False) -> torch.Tensor:
    """x [kp][>= n] 16-bit, row = contraction index (token) -> the fragment-major copy of x^T ([n][kp]): the B operand of
    :func:`gemm16_ta_fragw` (include/llark_hip.h: llark_pack_frag_t16; ``chunk16``: llark_pack_frag_t16x16, the 16-row chunks of the
    16x16x32 MFMA shape)."""
    kp = x.shape[0]
    n = x.shape[1] if n is None else n
    assert x.stride(1) == 1 and kp % 64 == 0 and n % 8 == 0
    out = torch.empty((round_up(n, 32) * kp,), dtype=x.dtype, device=x.device)
    fn = _lib.lib().llark_pack_frag_t16x16 if chunk16 else _lib.lib().llark_pack_frag_t16
    check(fn(_dev(x, "x", contiguous=False), x.stride(0), kp, n, _dev(out, "out"), _stream()), "pack_frag_t16")
    return out


def gemm16_ta_fragw_takes(m: int, kp: int, lda: int) -> bool:
    return m % 8 == 0 and kp % 64 == 0 and kp >= 192 and lda % 8 == 0 and kp * lda * 2 < (1 << 31)


def gemm16_ta_fragw16_takes(m: int, n: int, kp: int, lda: int, c: torch.Tensor) -> bool:
    """Whether llark_gemm16_ta_fragw16 (the 16x16x32 MFMA shape) takes the product into ``c`` (include/llark_hip.h)."""
    return (gemm16_ta_fragw_takes(m, kp, lda) and kp % 128 == 0 and kp >= 256 and n % 4 == 0 and c.stride(0) % 4 == 0
            and c.data_ptr() % 16 == 0)


def gemm16_ta_fragw(a: torch.Tensor, wfrag: torch.Tensor, m: int, n: int, kp: int, c: torch.Tensor, accumulate: bool = False,
                    sumsq: Optional[torch.Tensor] = None, chunk16: bool = False) -> None:
    """c[m][n] (= | +=) sum_k a[k][m] B(n, k): ``a`` [kp][>= m] bf16 contraction-major (dY as the backward leaves it), ``wfrag`` =
    :func:`pack_frag_t16` of X [kp][n] (include/llark_hip.h: llark_gemm16_ta_fragw; ``chunk16``: llark_gemm16_ta_fragw16 over
    ``pack_frag_t16(.., chunk16=True)``)."""
    bf = torch.bfloat16
    assert a.dtype == bf and wfrag.dtype == bf and a.stride(1) == 1 and wfrag.numel() == round_up(n, 32) * kp
    fn = _lib.lib().llark_gemm16_ta_fragw16 if chunk16 else _lib.lib().llark_gemm16_ta_fragw
    with _timed("gemm_bf16", 2.0 * m * n * kp):
        check(fn(
            EPI_RESID if accumulate else EPI_F32, _dev(a, "a", contiguous=False), a.stride(0), _dev(wfrag, "wfrag"), m, n, kp,
            _dev(c, "c", torch.float32, contiguous=False), c.stride(0), _dev(c, "c", torch.float32, contiguous=False) if accumulate else None,
            c.stride(0), _dev(sumsq, "sumsq", torch.float64) if sumsq is not None else None, _stream()), "gemm16_ta_fragw")


def gemm16_fragw_rope_qkv_train(a: torch.Tensor, wfrag: torch.Tensor, kp: int, batch: int, s: int, nh: int, pos0: int, cos_t: torch.Tensor,
                                sin_t: torch.Tensor, q: torch.Tensor, k_cache: torch.Tensor, vt_cache: torch.Tensor, v_rm: torch.Tensor) -> None:
    """:func:`gemm16_fragw_rope_qkv` (plain bf16 operands) that also writes V row-major, ``v_rm`` [batch][nh][s][128]: the attention
    backward's operand (include/llark_hip.h: llark_gemm16_fragw_rope_qkv_train)."""
    bf = torch.bfloat16
    hd = 128
    smax = k_cache.shape[-2]
    n = 3 * nh * hd
    assert a.dtype == bf and wfrag.dtype == bf and wfrag.numel() == n * kp and a.shape[0] >= batch * s and a.shape[1] >= kp
    assert k_cache.shape[-1] == hd and vt_cache.shape[-1] == smax and q.numel() >= batch * nh * s * hd and v_rm.numel() >= batch * nh * s * hd
    with _timed("gemm_bf16", 2.0 * batch * s * n * kp):
        check(_lib.lib().llark_gemm16_fragw_rope_qkv_train(
            _dev(a, "a", bf), a.stride(0), _dev(wfrag, "wfrag", bf), kp, batch, s, nh, hd, pos0, _dev(cos_t, "cos", torch.float32),
            _dev(sin_t, "sin", torch.float32), cos_t.shape[0], _dev(q, "q", bf), _dev(k_cache, "k_cache", bf), _dev(vt_cache, "vt_cache", bf),
            smax, _dev(v_rm, "v_rm", bf), _stream()), "gemm16_fragw_rope_qkv_train")


def attn_backward_fused(q, k_cache, v_rm, dO: torch.Tensor, o, lse, dsum, batch: int, s: int, nh: int, hd: int, cos_t: torch.Tensor,
                        sin_t: torch.Tensor, pos0: int, dqkv: torch.Tensor) -> None:
    """:func:`attn_backward` with the layout glue folded in (include/llark_hip.h: llark_attn_backward_bf16_fused): ``dO`` token-major
    [batch*s][>= nh*hd] (any row pitch), output ``dqkv`` bf16 [batch*s][3*nh*hd] = d(q | k | v) with the RoPE backward applied."""
    smax = k_cache.shape[-2]
    bf, f32 = torch.bfloat16, torch.float32
    assert dO.dtype == bf and dO.stride(1) == 1 and dO.shape[0] >= batch * s and dO.shape[1] >= nh * hd
    assert dqkv.dtype == bf and dqkv.is_contiguous() and dqkv.shape == (batch * s, 3 * nh * hd)
    check(_lib.lib().llark_attn_backward_bf16_fused(
        _dev(q, "q", bf), _dev(k_cache, "k_cache", bf), _dev(v_rm, "v_rm", bf), _dev(dO, "dO", bf, contiguous=False), dO.stride(0),
        _dev(o, "o", bf), _dev(lse, "lse", f32), _dev(dsum, "dsum", f32), batch, s, nh, hd, smax, _dev(cos_t, "cos", f32),
        _dev(sin_t, "sin", f32), pos0, cos_t.shape[0], _dev(dqkv, "dqkv", bf), _stream()), "attn_backward_fused")


# ---- decode-step launches chained across their boundaries (include/llark_hip.h: llark_gemv16_dma_chain) -----------------------------------
def gemv16_dma_blocks(epilogue: int, n: int) -> int:
    """Grid size of a streaming-Linear launch: what its arrival counter advances by per launch."""
    return int(_lib.lib().llark_gemv16_dma_blocks(int(epilogue), int(n)))


def gemv16_dma_chain(wt: torch.Tensor, n: int, epilogue: int, split: bool, a_hi=None, a_lo=None, x=None, norm_w=None, eps: float = 0.0,
                     c=None, resid=None, out_hi=None, out_lo=None, wait: Optional[torch.Tensor] = None, wait_target: int = 0,
                     signal: Optional[torch.Tensor] = None) -> None:
    """One-row streaming Linear as a link of a chain of overlapping launches: ``wait`` / ``signal`` are one-element int32 views of the
    caller's counter tensor.  ``x`` (fp32 row) + ``norm_w`` = the RMSNorm-fused form, else ``a_hi`` (+ ``a_lo``)."""
    bf, f32 = torch.bfloat16, torch.float32
    kp = wt.shape[1]
    name = ("gemm_split_" if split else "gemm_") + "bf16_skinny"
    with _timed(name, 2.0 * n * kp):
        check(_lib.lib().llark_gemv16_dma_chain(
            int(split), int(epilogue), _opt(a_hi, "a_hi", bf), _opt(a_lo, "a_lo", bf), a_hi.stride(0) if a_hi is not None else 0,
            _opt(x, "x", f32), x.stride(0) if x is not None else 0, _opt(norm_w, "norm_w", f32), float(eps), _dev(wt, "wt", bf), wt.stride(0),
            None, 1, n, kp, _opt(c, "c", f32), c.stride(0) if c is not None else 0, _opt(resid, "resid", f32),
            resid.stride(0) if resid is not None else 0, _opt(out_hi, "out_hi", bf), _opt(out_lo, "out_lo", bf),
            out_hi.stride(0) if out_hi is not None else 0, wait.data_ptr() if wait is not None else None, int(wait_target) & 0xFFFFFFFF,
            signal.data_ptr() if signal is not None else None, _stream()), "gemv16_dma_chain")


def attn_decode_rope_chain(qkv: torch.Tensor, batch: int, nh: int, hd: int, pos: int, cos_t, sin_t, k_cache, vt_cache, out,
                           k_cache_lo, vt_cache_lo, out_lo, done: torch.Tensor) -> None:
    """:func:`attn_decode_rope` (host position) whose workgroups add 1 to ``done`` when their heads are written (chained o_proj)."""
    smax = k_cache.shape[-2]
    bf = torch.bfloat16
    assert qkv.shape == (batch, 3 * nh * hd)
    check(_lib.lib().llark_attn_decode_rope_bf16_chain(
        _dev(qkv, "qkv", torch.float32), batch, nh, hd, int(pos), None, _dev(cos_t, "cos", torch.float32), _dev(sin_t, "sin", torch.float32),
        cos_t.shape[0], _dev(k_cache, "k_cache", bf), _dev(vt_cache, "vt_cache", bf), _opt(k_cache_lo, "k_cache_lo", bf),
        _opt(vt_cache_lo, "vt_cache_lo", bf), smax, _dev(out, "out", bf), _opt(out_lo, "out_lo", bf), None, done.data_ptr(), _stream()),
        "attn_decode_rope_chain")
