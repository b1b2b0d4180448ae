"""ctypes loader for ``libllark_hip.so`` (the C-ABI declared in ``include/llark_hip.h``).

The product path has NO CPU or PyTorch fallback: if the HIP library cannot be loaded this module
raises, and every op wrapper in :mod:`llark_amd.ops` goes through it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLARK_HIP_LIB") or os.path.join(_HERE, "libllark_hip.so")   # override: profiling builds only
CSRC = os.path.join(_HERE, "csrc")

_lib = None


class LlarkHipError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``llark_amd/libllark_hip.so`` (hipcc cross-compiles
    without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "llark_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs if os.path.exists(s)
    )
    if force or stale:
        cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
        res = subprocess.run(cmd, capture_output=not verbose, text=True)
        if res.returncode != 0:
            raise LlarkHipError(f"building libllark_hip.so failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


# name -> argtypes (all return int unless noted)
_P = c_void_p
_SIGS = {
    "llark_version": [],
    "llark_device_info": [c_int, c_char_p, c_int],
    "llark_resample_sinc_host": [_P, c_int64, c_double, _P, _P, c_int, c_int, _P, c_int64],
    "llark_flac_info_host": [_P, c_int64, _P, _P, _P, _P],
    "llark_flac_decode_host": [_P, c_int64, _P, c_int64, _P, c_int],
    "llark_pack_conv_weight": [_P, _P, c_int, c_int, c_int, _P],
    "llark_conv1d_f32": [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "llark_resblock_f32": [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P],
    "llark_codebook_norms_f32": [_P, c_int, c_int, _P, _P],
    "llark_codebook_argmin": [_P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P],
    "llark_prior_embed": [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P],
    "llark_layernorm_split_f16": [_P, c_int, c_int, c_int, _P, _P, c_float, _P, _P, c_int, _P],
    "llark_prior_attn": [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P],
    "llark_pool_window": [_P, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "llark_pool_mean": [_P, c_int, c_int, c_int, _P, _P, _P],
    "llark_zero_pad16": [_P, c_int, c_int, c_int, _P],
    "llark_vqvae_plan_add_conv": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int],
    "llark_vqvae_plan_add_resblock": [_P, _P, _P, _P, _P, c_int, c_int],
    "llark_vqvae_encode": [_P, _P, c_int, c_int, _P, _P, c_int64, _P, _P, c_int, _P, _P, _P],
    "llark_codebook_argmin_tie": [_P, c_int, c_int, c_int, _P, _P, c_int, _P, c_float, c_float, _P, _P, c_int, _P],
    "llark_vqvae_fix_near_ties": [_P, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, _P, _P, c_int, _P, _P],
    "llark_vqvae_stage_f16x2": [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "llark_vqvae_pack_frag16": [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P],
    "llark_gemm16": [c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int,
                     _P, _P, c_int, _P],
    "llark_gemm16_ex": [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int,
                        _P, _P, c_int, _P],
    "llark_gemm16_ws": [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int,
                        _P, _P, c_int, _P, _P],
    "llark_gemm16_lo8": [c_int, _P, _P, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int,
                         _P, _P, c_int, c_int, _P, _P],
    "llark_pack_weight_lo8": [_P, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "llark_gemm16_ln_takes": [c_int, c_int, c_int],
    "llark_gemm16_ln": [c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P],
    "llark_ln_stats_finalize": [_P, c_int, c_int, c_int, c_float, _P, _P],
    "llark_gemm16_fragw_whole_tiles": [c_int, c_int, c_int, c_int, c_int],
    "llark_gemm16_ln_p": [c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P],
    "llark_gemm16_lnp_fragw": [c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P, _P, _P, _P],
    "llark_ln_stats_finalize_p": [_P, c_int, c_int, c_int, c_float, _P, _P, _P],
    "llark_ln_row_pred": [_P, c_int, c_int, c_int, c_float, _P, _P],
    "llark_workspace_destroy": [_P],
    "llark_layernorm_split_lo8": [_P, c_int, c_int, c_int, _P, _P, c_float, _P, c_int, _P, c_int, c_int, _P],
    "llark_prior_attn_lo8": [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, _P],
    "llark_pack_weight16_frag": [_P, c_int, c_int, c_int, _P, _P],
    "llark_gemm16_fragw": [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int,
                           _P, _P, c_int, _P],
    "llark_gemm16_fragw_sk": [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int,
                              _P, _P, c_int, _P, c_int64, _P],
    "llark_gemm16_fragw_rope_qkv": [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P],
    "llark_gemm16_resid_rmsnorm": [c_int, c_int, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_float, _P, _P,
                                   c_int, _P],
    "llark_gemm16_rmsnorm_a": [c_int, c_int, c_int, _P, c_int, _P, c_float, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, _P,
                               c_int, _P],
    "llark_gemm16_batched": [c_int, c_int, c_int, _P, _P, c_int, c_int64, _P, c_int, c_int64, c_int, c_int, c_int, _P, c_int,
                             c_int64, _P, _P, c_int, c_int64, c_int, _P],
    "llark_pack_weight16": [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P],
    "llark_split16": [c_int, _P, c_int, c_int, c_int, _P, _P, c_int, _P],
    "llark_embed_gather": [_P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P],
    "llark_rmsnorm_bf16": [_P, c_int, c_int, c_int, _P, c_float, _P, _P, c_int, _P],
    "llark_attn_prefill_bf16_alibi": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "llark_attn_decode_bf16_alibi": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "llark_gemv16_dma": [c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P],
    "llark_gemv16_dma_rmsnorm": [c_int, c_int, _P, c_int, _P, c_float, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P],
    "llark_gemm16_t": [c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P],
    "llark_gemm16_t_ex": [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P],
    "llark_gemm16_t_sumsq": [c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P],
    "llark_attn_prefill_bf16_lse": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "llark_attn_backward_bf16": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P],
    "llark_layernorm_bf16": [_P, c_int, c_int, c_int, _P, _P, c_float, _P, _P, c_int, _P],
    "llark_layernorm_f32": [_P, c_int, c_int, c_int, _P, _P, c_float, _P, c_int, _P],
    "llark_clamp_f32": [_P, c_int64, c_float, _P],
    "llark_clamp_bwd_bf16": [_P, c_int64, c_float, _P, _P],
    "llark_sumsq_f32": [_P, c_int64, _P, c_int, _P],
    "llark_scale_f32": [_P, c_int64, c_float, _P],
    "llark_gelu_split_bf16": [_P, c_int, c_int, c_int, _P, _P, c_int, _P],
    "llark_layernorm_bwd": [_P, c_int, _P, _P, c_int, c_int, c_int, c_float, _P, c_int, _P, _P, c_int, _P],
    "llark_gelu_bwd": [_P, _P, c_int64, _P, _P, _P],
    "llark_causal_softmax_rows_alibi": [_P, c_int, c_int, c_float, _P, c_int, _P, c_int, _P],
    "llark_clap_logmel": [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P],
    "llark_clap_patchify": [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, c_int, _P],
    "llark_clap_window_attn": [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P],
    "llark_layernorm_bf16_dup": [_P, c_int, c_int, c_int, _P, _P, c_float, _P, _P, _P, c_int, _P],
    "llark_gemm16_act": [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P],
    "llark_clap_patch_merge": [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "llark_mean_rows_f32": [_P, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "llark_relu_split_bf16": [_P, c_int, c_int, c_int, _P, _P, c_int, _P],
    "llark_l2_normalize_rows": [_P, c_int, c_int, c_int, c_float, _P],
    "llark_attn_decode_rope_bf16": [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P],
    "llark_rope_split_heads_dpos": [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P],
    "llark_attn_decode_bf16_dpos": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, _P, _P],
    "llark_rope_split_heads": [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P],
    "llark_attn_prefill_bf16": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P],
    "llark_attn_decode_bf16": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P],
    "llark_cross_entropy_shifted": [_P, c_int, c_int, c_int, c_int, _P, c_int64, _P, _P, _P],
    "llark_transpose16": [_P, c_int, c_int, c_int, _P, c_int, c_int, c_int64, c_int64, _P],
    "llark_split_heads16": [_P, c_int, c_int, c_int, c_int, _P, _P],
    "llark_causal_softmax_rows": [_P, c_int, c_int, c_float, _P, c_int, _P],
    "llark_attn_ds": [_P, _P, c_int, c_int, c_float, _P, c_int, _P],
    "llark_rope_merge_bwd": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P],
    "llark_rmsnorm_bwd": [_P, _P, _P, c_int, c_int, c_float, _P, c_int, _P, _P],
    "llark_rmsnorm_bwd_out16": [_P, _P, _P, c_int, c_int, c_float, _P, c_int, _P, _P, c_int, _P],
    "llark_swiglu_fwd": [_P, c_int, c_int, _P, _P],
    "llark_swiglu_bwd": [_P, _P, c_int, c_int, _P, _P],
    "llark_cross_entropy_bwd": [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, _P, c_int, _P],
    "llark_colsum_f32": [_P, c_int, c_int, c_int, _P, _P],
    "llark_gather_rows_f32": [_P, c_int, _P, c_int, c_int, _P, c_int, _P],
    "llark_scatter_add_rows_f32": [_P, c_int, _P, c_int, c_int, _P, c_int, _P],
    "llark_adamw": [c_int, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, _P],
    "llark_adamw_clip": [c_int, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float, _P, c_float, _P],
    "llark_gemm16_fragw_swiglu_train": [c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P],
    "llark_pack_frag_t16": [_P, c_int, c_int, c_int, _P, _P],
    "llark_gemm16_ta_fragw": [c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P],
    "llark_pack_frag_t16x16": [_P, c_int, c_int, c_int, _P, _P],
    "llark_gemm16_ta_fragw16": [c_int, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P],
    "llark_gemm16_fragw_rope_qkv_train": [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, c_int, _P, _P],
    "llark_attn_backward_bf16_fused": [_P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, _P],
    "llark_gemv16_dma_blocks": [c_int, c_int],
    "llark_gemv16_dma_chain": [c_int, c_int, _P, _P, c_int, _P, c_int, _P, c_float, _P, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P, c_uint, _P, _P],
    "llark_attn_decode_rope_bf16_chain": [_P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P],
    "llark_adamw_twins": [_P, _P, _P, _P, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_float, _P, c_float, _P, c_int, _P, _P],
}


def declared_symbols():
    """Every symbol ``include/llark_hip.h`` declares (kept in sync by tests/test_abi.py)."""
    return sorted(list(_SIGS.keys()) + ["llark_last_error", "llark_vqvae_plan_create", "llark_vqvae_plan_destroy", "llark_workspace_create", "llark_gemm16_sk_scratch_bytes"])


# ---- host-side launch lists -------------------------------------------------------------------------------------------
# A fixed-shape forward is the same sequence of C-ABI calls with the same pointers every time.  While a recorder is
# installed, lib() hands out a proxy that executes each call AND appends (function, name, args, timing label) to the
# recorder; ops.LaunchList.replay() then re-issues the list without re-deriving shapes, views or pointers in Python
# (the per-launch host cost drops from tens of microseconds to the ctypes call itself).  hipGraph replay of the same
# sequences measured slower than eager launches on ROCm 7.2 (profiles/r01_gemm_ablation.txt), hence a host list.
_recorder = None
current_label = None          # (name, work) of the ops._timed region the next call belongs to


class _RecordingProxy:
    def __init__(self, real, rec):
        self._real, self._rec = real, rec

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("llark_") or name == "llark_last_error":
            return fn
        rec = self._rec

        def call(*args):
            rc = fn(*args)
            rec.append((fn, name, args, current_label))
            return rc

        return call


def set_recorder(rec) -> None:
    global _recorder
    _recorder = rec


def lib():
    L = _real_lib()
    return _RecordingProxy(L, _recorder) if _recorder is not None else L


def _real_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # same image on the GPU box: hipcc is present, so a missing .so is built, never skipped
            build()
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # fail loudly: there is no fallback path
            raise LlarkHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_int
        L.llark_last_error.argtypes = []
        L.llark_last_error.restype = c_char_p
        L.llark_vqvae_plan_create.argtypes = []
        L.llark_vqvae_plan_create.restype = c_void_p
        L.llark_vqvae_plan_destroy.argtypes = [c_void_p]
        L.llark_vqvae_plan_destroy.restype = None
        L.llark_workspace_create.argtypes = []
        L.llark_workspace_create.restype = c_void_p
        L.llark_gemm16_sk_scratch_bytes.argtypes = []
        L.llark_gemm16_sk_scratch_bytes.restype = c_int64
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().llark_last_error().decode("utf-8", "replace")
        raise LlarkHipError(f"{what or 'llark_hip'} failed (code {rc}): {msg}")
