"""Hyper-parameters reached by the reference's ``load_model("5b")`` (jukebox/main.py:176-200).

Upstream names (openai/jukebox @ 08efbbc ``hparams.py``: ``vqvae`` + ``prior_5b`` with
``prior_depth`` forced to 36 at jukebox/main.py:198).  Only the fields the embedding path reads
are kept.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple


@dataclass
class JukeboxHParams:
    sr: int = 44100
    n_samples: int = 8                     # jukebox/main.py:187 (conditioning batch; only [0] is kept)
    sample_length: int = 1048576           # jukebox/main.py:194
    # vqvae, level-2 (top) encoder
    downs_t: Tuple[int, ...] = (3, 2, 2)
    strides_t: Tuple[int, ...] = (2, 2, 2)
    emb_width: int = 64
    l_bins: int = 2048
    width: int = 32
    depth: int = 4
    dilation_growth_rate: int = 3
    # prior_5b
    n_ctx: int = 8192
    prior_width: int = 4800
    prior_depth: int = 36                  # jukebox/main.py:198
    heads: int = 8
    blocks: int = 128
    m_attn: float = 0.25
    m_mlp: float = 1.0
    init_scale: float = 0.1
    y_bins: Tuple[int, int] = (604, 7898)
    t_bins: int = 128
    max_bow_genre_size: int = 5
    min_duration: float = 60.0
    max_duration: float = 600.0
    cond_seconds: int = 62                 # jukebox/main.py:72

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
    def head_dim(self) -> int:
        return self.n_state // self.heads

    @property
    def block_ctx(self) -> int:
        return self.n_ctx // self.blocks

    @property
    def mlp_state(self) -> int:
        return int(self.m_mlp * self.prior_width)

    def check(self) -> None:
        assert self.sample_length == self.n_ctx * self.raw_to_tokens, "sample_length != n_ctx*raw_to_tokens"
        assert self.n_state % self.heads == 0
        assert self.n_ctx % self.blocks == 0
        assert self.prior_width % 4 == 0


def hparams_5b() -> JukeboxHParams:
    return JukeboxHParams()


def hparams_tiny() -> JukeboxHParams:
    """Structure-preserving small twin (3 level blocks, all 3 attention patterns) for fast tests."""
    return JukeboxHParams(sample_length=65536, n_ctx=512, prior_width=192, prior_depth=3, heads=2, blocks=8)


def hparams_5b_depth(depth: int) -> JukeboxHParams:
    """Full 5b widths with a reduced depth (parity tests that must finish in seconds on CPU)."""
    return JukeboxHParams(prior_depth=depth)
