"""ctypes binding of oracle/jukebox_ref.c (CPU ORACLE -- test infrastructure, not product code).

Drives the bit-exact C restatement layer by layer over the level-2 encoder of the VQ-VAE
(jukebox/main.py:61 -> upstream VQVAE.encode) and the codebook search.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libjukebox_ref.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "jukebox_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        L.jbref_conv1d.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int]
        L.jbref_resblock.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, ctypes.c_int, fp]
        L.jbref_codebook.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int64), fp]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _np(t) -> np.ndarray:
    return np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)


def conv1d(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride=1, pad=0, dil=1, relu_in=False) -> np.ndarray:
    cin, tin = x.shape
    cout, cin2, k = w.shape
    assert cin == cin2
    tout = (tin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y = np.empty((cout, tout), dtype=np.float32)
    lib().jbref_conv1d(_p(x), cin, tin, _p(w), _p(b), cout, k, stride, pad, dil, int(relu_in), _p(y), tout)
    return y


def resblock(x, w1, b1, w2, b2, dil) -> np.ndarray:
    c, t = x.shape
    y = np.empty_like(x)
    lib().jbref_resblock(_p(x), c, t, _p(w1), _p(b1), _p(w2), _p(b2), dil, _p(y))
    return y


def codebook(x: np.ndarray, k: np.ndarray):
    emb, t = x.shape
    bins = k.shape[0]
    codes = np.empty((t,), dtype=np.int64)
    mind = np.empty((t,), dtype=np.float32)
    lib().jbref_codebook(_p(x), emb, t, _p(k), bins, codes.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _p(mind))
    return codes, mind


def encoder_forward(w: Dict[str, object], x: np.ndarray, spec, level: int = 2,
                    taps: Optional[List[np.ndarray]] = None) -> np.ndarray:
    """x: [1][T] one clip (channel-major) -> [emb][T/raw_to_tokens]. Same layer walk as
    oracle/jukebox_ref.py::vqvae_encoder_forward; `taps` collects every layer output."""
    p = f"encoders.{level}"
    x = np.ascontiguousarray(x, dtype=np.float32)
    for lb, (down_t, stride_t) in enumerate(zip(spec.downs_t, spec.strides_t)):
        for i in range(down_t):
            b = f"{p}.level_blocks.{lb}.model.{i}"
            x = conv1d(x, _np(w[f"{b}.0.weight"]), _np(w[f"{b}.0.bias"]), stride=stride_t, pad=stride_t // 2)
            if taps is not None:
                taps.append(x)
            for r in range(spec.depth):
                d = spec.dilation_growth_rate ** r
                rb = f"{b}.1.model.{r}.model"
                x = resblock(x, _np(w[f"{rb}.1.weight"]), _np(w[f"{rb}.1.bias"]),
                             _np(w[f"{rb}.3.weight"]), _np(w[f"{rb}.3.bias"]), d)
                if taps is not None:
                    taps.append(x)
        b = f"{p}.level_blocks.{lb}.model.{down_t}"
        x = conv1d(x, _np(w[f"{b}.weight"]), _np(w[f"{b}.bias"]), stride=1, pad=1)
        if taps is not None:
            taps.append(x)
    return x


def encode_codes(w: Dict[str, object], audio: np.ndarray, spec, return_all: bool = False):
    """audio: (N, sample_length) fp32 -> codes (N, n_ctx) int64 (bit-exact oracle)."""
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    if audio.ndim == 1:
        audio = audio[None]
    k = _np(w["bottleneck.level_blocks.2.k"])
    out, encs, minds = [], [], []
    for n in range(audio.shape[0]):
        xe = encoder_forward(w, audio[n][None, : spec.sample_length], spec)
        c, md = codebook(xe, k)
        out.append(c)
        encs.append(xe)
        minds.append(md)
    if return_all:
        return np.stack(out), np.stack(encs), np.stack(minds)
    return np.stack(out)
