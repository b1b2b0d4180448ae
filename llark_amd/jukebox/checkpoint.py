"""Reader for upstream Jukebox ``.pth.tar`` checkpoints -- the on-disk door of ``load_model`` (jukebox/main.py:176-200).

What the reference does (through upstream ``jukebox/make_models.py`` @ 08efbbc, patched by
``/root/reference/jukebox/make_models.py.patch:7-8``):

* ``make_vqvae`` / ``make_prior`` call ``restore_model(hps, model, hps.restore_vqvae | hps.restore_prior)``; the paths are
  ``REMOTE_PREFIX + "jukebox/models/5b/vqvae.pth.tar"`` and ``".../5b/prior_level_2.pth.tar"`` (``5b_lyrics/`` for the
  lyrics model), mirrored under ``~/.cache/jukebox/models/`` by ``load_checkpoint``;
* the file is ``torch.save({"model": state_dict, "step": ..., "hps": ...})``; keys may carry DDP's ``module.`` prefix,
  which ``restore_model`` strips;
* the patch turns ``load_state_dict(checkpoint["model"])`` into ``strict=False`` because jukebox/main.py:196 builds the
  prior with ``prior_depth = 36`` while the 5b checkpoint holds 72 ``_attn_mods`` -- layers >= 36 (and anything else the
  module tree does not have) are dropped without complaint.

Here the model is not a module tree but a name -> tensor mapping consumed by ``VQVAE`` / ``TopPrior``: the same filter is
applied to the names.  One deliberate difference: ``strict=False`` would also leave MISSING parameters at their random
initial values without a word; a checkpoint that lacks a tensor the path needs raises ``KeyError`` here instead.
"""
from __future__ import annotations

import logging
import os
import pickle
import re
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .hparams import JukeboxHParams

CACHE_ENV = "JUKEBOX_CACHE"                 # default: ~/.cache  (upstream load_checkpoint: ~/.cache/<remote path>)
_ATTN_MOD = re.compile(r"^prior\.transformer\._attn_mods\.(\d+)\.")


def default_checkpoint_paths(model: str = "5b") -> Tuple[str, str]:
    """Local mirror paths upstream ``load_checkpoint`` uses for the 5b / 5b_lyrics VQ-VAE and top-level prior."""
    if model not in ("5b", "5b_lyrics"):
        raise ValueError(f"unknown model {model!r}")
    root = os.path.join(os.environ.get(CACHE_ENV, os.path.expanduser("~/.cache")), "jukebox", "models")
    return os.path.join(root, "5b", "vqvae.pth.tar"), os.path.join(root, model, "prior_level_2.pth.tar")


UNSAFE_PICKLE_ENV = "LLARK_ALLOW_UNSAFE_CHECKPOINT_PICKLE"


def read_pth_tar(path, allow_pickle: Optional[bool] = None) -> Dict[str, torch.Tensor]:
    """``checkpoint["model"]`` of an upstream checkpoint file with DDP's ``module.`` prefix removed
    (upstream make_models.py ``restore_model``).  Accepts a path or a file object.

    The file is read with ``torch.load(weights_only=True)``: upstream's files are a dict of tensors plus an ``hps`` dict of
    plain Python values, which the restricted unpickler accepts.  A file it rejects is NOT retried with the full
    unpickler (that would run whatever code a corrupted or hostile ``.pth.tar`` carries) unless the caller opts in with
    ``allow_pickle=True`` or ``LLARK_ALLOW_UNSAFE_CHECKPOINT_PICKLE=1`` -- the same gate m2t/data.py puts on ``.pyd``."""
    if allow_pickle is None:
        allow_pickle = os.environ.get(UNSAFE_PICKLE_ENV, "0") == "1"
    pos = path.tell() if hasattr(path, "tell") else None
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not allow_pickle:
            raise ValueError(f"{path}: the restricted (weights_only) loader rejected this checkpoint: {e}.  If the file is trusted, "
                             f"pass allow_pickle=True or set {UNSAFE_PICKLE_ENV}=1 to read it with the full unpickler.") from e
        logging.warning("%s: reading with the UNRESTRICTED unpickler (%s)", path, UNSAFE_PICKLE_ENV)
        if pos is not None:
            path.seek(pos)
        ck = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(ck, dict) or "model" not in ck:
        raise ValueError(f"{path}: not a Jukebox checkpoint (expected a dict with a 'model' entry)")
    return {(k[7:] if k[:7] == "module." else k): v for k, v in ck["model"].items()}


def vqvae_names(hps: JukeboxHParams) -> List[str]:
    """State-dict names of the level-2 encoder + codebook: what ``VQVAE`` consumes (upstream Encoder / EncoderConvBlock /
    Resnet1D / ResConv1DBlock module paths)."""
    names = []
    p = "encoders.2"
    for lb, down_t in enumerate(hps.downs_t):
        for i in range(down_t):
            b = f"{p}.level_blocks.{lb}.model.{i}"
            names += [f"{b}.0.weight", f"{b}.0.bias"]
            for r in range(hps.depth):
                rb = f"{b}.1.model.{r}.model"
                names += [f"{rb}.1.weight", f"{rb}.1.bias", f"{rb}.3.weight", f"{rb}.3.bias"]
        b = f"{p}.level_blocks.{lb}.model.{down_t}"
        names += [f"{b}.weight", f"{b}.bias"]
    names.append("bottleneck.level_blocks.2.k")
    return names


def prior_names(hps: JukeboxHParams, depth: Optional[int] = None) -> List[str]:
    """State-dict names of the top-level prior in ``only_encode`` mode: embeddings, ``depth`` ResAttnBlocks and the label
    conditioner tables (``x_out`` and ``start_token`` are never reached on this path)."""
    depth = hps.prior_depth if depth is None else depth
    names = ["prior.x_emb.weight", "prior.pos_emb.pos_emb"]
    for d in range(depth):
        p = f"prior.transformer._attn_mods.{d}"
        names += [f"{p}.ln_0.weight", f"{p}.ln_0.bias", f"{p}.ln_1.weight", f"{p}.ln_1.bias"]
        for m in ("attn.c_attn", "attn.c_proj", "mlp.c_fc", "mlp.c_proj"):
            names += [f"{p}.{m}.w", f"{p}.{m}.b"]
    names += [f"y_emb.{n}.emb.weight" for n in ("bow_genre_emb", "artist_emb", "total_length_emb", "absolute_pos_emb",
                                                "relative_pos_emb")]
    return names


def select_weights(state_dicts: Iterable[Dict[str, torch.Tensor]], hps: JukeboxHParams, depth: Optional[int] = None,
                   origin: str = "checkpoint"):
    """``load_state_dict(..., strict=False)`` over name -> tensor mappings: keeps exactly the tensors the path consumes,
    returns ``(weights, unexpected)`` where ``unexpected`` lists what was dropped (layers >= depth, decoders, the other
    levels, ``prior.x_out`` ...).  Missing tensors raise ``KeyError`` (see the module docstring)."""
    merged: Dict[str, torch.Tensor] = {}
    for sd in state_dicts:
        merged.update(sd)
    wanted = vqvae_names(hps) + prior_names(hps, depth)
    missing = [n for n in wanted if n not in merged]
    if missing:
        raise KeyError(f"{origin} lacks {len(missing)} tensors the Jukebox path needs, e.g. {missing[:4]}")
    want = set(wanted)
    unexpected = sorted(k for k in merged if k not in want)
    d_eff = hps.prior_depth if depth is None else depth
    dropped_layers = sorted({int(m.group(1)) for k in unexpected for m in [_ATTN_MOD.match(k)] if m})
    if dropped_layers:
        logging.info("%s: prior layers %d..%d dropped (prior_depth = %d, jukebox/main.py:196 + make_models.py.patch)",
                     origin, dropped_layers[0], dropped_layers[-1], d_eff)
    return {n: merged[n] for n in wanted}, unexpected


def load_checkpoint_weights(model: str, hps: JukeboxHParams, depth: Optional[int] = None,
                            restore_vqvae: Optional[str] = None, restore_prior: Optional[str] = None):
    """Reads the two checkpoint files of ``model`` and returns ``(weights, unexpected)``; raises ``FileNotFoundError``
    naming the paths when a file is absent (the reference would download it; there is no network path here)."""
    dv, dp = default_checkpoint_paths(model)
    pv, pp = restore_vqvae or dv, restore_prior or dp
    absent = [p for p in (pv, pp) if not os.path.exists(p)]
    if absent:
        raise FileNotFoundError(
            f"Jukebox checkpoint file(s) not found: {absent}.  Place upstream's vqvae.pth.tar / prior_level_2.pth.tar there "
            f"(or set ${CACHE_ENV}, or pass restore_vqvae= / restore_prior=), or ask for seeded synthetic weights "
            f"explicitly (load_model(..., weights='synthetic') / --synthetic-weights).")
    return select_weights([read_pth_tar(pv), read_pth_tar(pp)], hps, depth, origin=f"{pv} + {pp}")
