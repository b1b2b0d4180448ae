"""Shared recipe of the FULL-DEPTH parity fixtures: the configuration bench.py times --
36 prior layers at 5b widths, 32 Llama-2-7B layers at S = 371 -- evaluated ONCE by the CPU oracle in the build
container (tests/golden/make_jukebox_full_golden.py, tests/golden/make_llama7b_golden.py) and re-evaluated by the HIP
path on the GPU box (tests/test_fulldepth_gpu.py).  Both sides regenerate the weights from the same CPU seeds; only
the oracle's outputs are committed (tests/golden/jukebox_full36.npz, tests/golden/llama7b_full32.npz).

Test infrastructure: imports oracle/, never imported by llark_amd/.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
JUKEBOX_NPZ = os.path.join(GOLDEN, "jukebox_full36.npz")
LLAMA_NPZ = os.path.join(GOLDEN, "llama7b_full32.npz")
WIDE_NPZ = os.path.join(GOLDEN, "jukebox_full36_wide.npz")     # round 4: more clips, a short clip, outlier weights, code table

CAL_CLIP = 100000            # bench.py's calibration clip for the data-dependent codebook
GOLD_CLIP = 0                # the clip whose codes / embedding are pinned
PROBE_ROWS = (0, 1, 127, 128, 4095, 4096, 8190, 8191)     # un-pooled activation rows kept in the fixture
PROBE_LAYERS = (1, 3, 6, 12, 24, 36)                      # depth at which the probe rows are recorded


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------------------------------------------------------------------------------------
# Jukebox half
# ---------------------------------------------------------------------------------------------
def jukebox_hps():
    from llark_amd.jukebox.hparams import hparams_5b

    return hparams_5b()


def jukebox_clip(i: int, hps) -> np.ndarray:
    """Exactly bench.py's clip(i): 25 s synthetic, peak-normalised, truncated to sample_length."""
    from llark_amd.jukebox.synthetic import synthetic_clip
    from oracle import jukebox_ref as R

    a = R.normalize_audio(synthetic_clip(i, seconds=25.0))
    a = np.pad(a, (0, max(0, hps.sample_length - len(a))))[: hps.sample_length]
    return a.astype(np.float32)


# --- round 4: the wide fixture (VERDICT r03 items 1 + 2) -------------------------------------------------------
CODE_CLIPS = tuple(range(10))       # bench.py's 8 clips (rank 0: clip_indices(0, 8) = 0..7) + 2 more: C-oracle code table
WIDE_CASES = {                      # name -> (clip kind, clip index, seconds, outlier weights?)
    "rich": ("rich", 0, 25.0, False),        # a second spectrum (synthetic_clip_rich), full length
    "short12": ("plain", 3, 12.0, False),    # 12 s: latent_audio_len = 4134 < 8192 (jukebox/main.py:136,154), 121 frames
    "outlier": ("plain", 0, 25.0, True),     # clip 0 through weights with x30 outlier channels in every prior layer
}
HEAD_TOKENS = 1024                  # the float64 noise-floor reference covers the first 1024 tokens (16 blocks; every attention
                                    # pattern of the prior is causal, so a prefix is self-contained)
OUTLIER_CHANNELS, OUTLIER_FACTOR = 4, 30.0


def jukebox_case_audio(kind: str, idx: int, seconds: float):
    """Peak-normalised waveform of a wide-fixture case BEFORE padding (len < sample_length for the short case)."""
    from llark_amd.jukebox.synthetic import synthetic_clip, synthetic_clip_rich
    from oracle import jukebox_ref as R

    gen = synthetic_clip_rich if kind == "rich" else synthetic_clip
    return R.normalize_audio(gen(idx, seconds=seconds)).astype(np.float32)


def add_outlier_channels(w, hps, n_ch: int = OUTLIER_CHANNELS, factor: float = OUTLIER_FACTOR, seed: int = 11):
    """In place: the SAME `n_ch` hidden channels become outliers in every prior layer, the way trained transformers carry a few
    persistent massive dimensions: the residual-writing columns of attn.c_proj / mlp.c_proj x factor on those channels (the
    residual stream then holds |h| ~ 100 there next to O(1) elsewhere).  Values stay fp16-representable (Conv1D.w is fp16-valued
    upstream).  (A first version also multiplied the LayerNorm gains: max|h| ~ 6000 and the fp32 oracle itself 7e2 away from
    float64 -- a chaotic system, useless as a parity case.)"""
    g = torch.Generator().manual_seed(seed)
    ch = torch.randperm(hps.prior_width, generator=g)[:n_ch]
    d = 0
    while f"prior.transformer._attn_mods.{d}.ln_0.weight" in w:
        p = f"prior.transformer._attn_mods.{d}"
        for name in ("attn.c_proj.w", "mlp.c_proj.w"):
            t = w[f"{p}.{name}"]
            t[:, ch] = (t[:, ch].float() * factor).to(t.dtype)
        d += 1
    return [int(c) for c in ch]


def jukebox_weights_cpu(hps, depth=None):
    """Seeded CPU-generator weights (the bench uses the device generator for speed; parity needs values both sides
    can regenerate)."""
    from llark_amd.jukebox.synthetic import make_jukebox_weights

    return make_jukebox_weights(hps, seed=0, depth=depth, device="cpu")


def codebook_from_encoding(xe, hps) -> torch.Tensor:
    from llark_amd.jukebox.synthetic import init_codebook_from_encodings

    return init_codebook_from_encodings(torch.as_tensor(xe), hps.l_bins)


# ---------------------------------------------------------------------------------------------
# Llama half
# ---------------------------------------------------------------------------------------------
def llama_spec(layers: int = 32):
    from llark_amd.m2t import bench_support as BS
    from oracle import llama_ref as LR

    return LR.LlamaSpec(num_hidden_layers=layers, vocab_size=BS.VOCAB, audio_start_token=BS.START,
                        audio_end_token=BS.END, audio_patch_token=BS.PATCH)


def llama_weights_cpu(spec):
    """bf16-valued weights (what the reference's model.to(bfloat16) holds), HF names, CPU seed 0."""
    from oracle import llama_ref as LR

    return LR.make_weights(spec, seed=0, std=0.02, dtype=torch.bfloat16)


def llama_inputs(batch: int = 1):
    """bench.py's prompt layout (S = 371) + N(0,1) audio frames standing in for encoder output (seed 3)."""
    from llark_amd.m2t import bench_support as BS

    ids = BS.make_prompt_ids(batch)
    g = torch.Generator().manual_seed(3)
    aud = torch.randn(batch, BS.FRAMES, 4800, generator=g)
    return ids, aud
