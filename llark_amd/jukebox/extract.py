"""Drop-in mirror of the reference's ``jukebox/main.py`` function set running on the MI355X HIP kernels.

Same function names, argument meaning and error behaviour as the reference:

  load_audio_from_file   jukebox/main.py:29-45      maybe_pad_audio_to_max_len  :48-51
  get_z                  :54-68                      get_cond                    :71-98
  get_final_activations  :101-110                    windowed_average            :113-130
  get_acts_from_file     :133-173                    load_model                  :176-200
  main                   :203-254

plus two additions that do not exist in the reference (documented as new in DESIGN.md):
``get_acts_from_audio_batch`` (a batched ``get_acts_from_file`` over in-memory waveforms, B clips
per launch instead of one file per iteration) and :class:`WrappedAudioEncoder` (an ``nn.Module``
whose ``forward(audio) -> (B, frames, 4800)`` wraps the get_z -> get_cond ->
get_final_activations -> pool sequence; the name comes from BASELINE.json's north_star).
"""
from __future__ import annotations

import logging
import os
from math import floor
from typing import Optional, Sequence

import numpy as np
import torch

from .. import ops
from .hparams import JukeboxHParams, hparams_5b
from .prior import TopPrior
from .vqvae import VQVAE

JUKEBOX_SAMPLE_RATE = 44100
T = 8192  # time dimension of jukebox activations

JUKEBOX_EXPECTED_SAMPLES_LEN = 1048576
JUKEBOX_SAMPLE_SECONDS = JUKEBOX_EXPECTED_SAMPLES_LEN / JUKEBOX_SAMPLE_RATE
ACTS_SAMPLE_RATE = T / JUKEBOX_SAMPLE_SECONDS


class EmptyFileError(ValueError):
    pass


def _normalize(audio: np.ndarray) -> np.ndarray:
    """jukebox/main.py:36-45: mono mean + peak normalisation."""
    if audio.ndim == 1:
        audio = audio[np.newaxis]
    audio = audio.mean(axis=0)
    norm_factor = np.abs(audio).max()
    if norm_factor > 0:
        audio = audio / norm_factor
    return audio.flatten()


def load_audio_from_file(fpath, res_type: Optional[str] = None) -> np.ndarray:
    """Reads a wav / FLAC / AIFF / .au file, resamples to 44.1 kHz, converts to mono, peak-normalises.

    ``fpath``: a path or a binary file object (the Beam worker passes ``io.BytesIO(wav_bytes)``,
    jukebox/dataflow_inference.py:101-103).  The reference calls ``librosa.load(fpath, sr=44100)`` (librosa / soundfile /
    resampy are not installed here): wav decoding uses ``scipy.io.wavfile`` with soundfile's integer scaling (int16 / 2^15,
    int32 and 24-bit-in-int32 / 2^31, uint8 -> (x - 128) / 2^7), FLAC a native decoder checked against the stream's own MD5
    signature (:mod:`llark_amd.jukebox.audio_decode`), mono = channel mean BEFORE resampling like
    ``librosa.load``.  Resampling (``res_type``, default ``$LLARK_RES_TYPE`` or "kaiser_best"): the band-limited sinc
    interpolation of the librosa 0.7.2 / resampy pair the reference's image installs, or "soxr_hq" for newer librosa's default
    -- both restated in :mod:`llark_amd.jukebox.resample`, neither pinnable offline; 44.1 kHz files are bit-identical.
    The reference's pipeline names its files wav on both sides (``read_wav_bytes``, ``input_filename.replace(".wav", ".npy")``,
    jukebox/main.py:251); what libsndfile would also read without further codecs -- FLAC, AIFF / AIFF-C, .au -- is decoded,
    mp3 / ogg (librosa's audioread fallback) are not.
    """
    from .audio_decode import decode_audio
    from .resample import resample

    try:
        sr, data = decode_audio(fpath)
        if data.size == 0:
            raise ValueError("empty file")
    except ValueError as ve:
        raise EmptyFileError(f"file {fpath} failed to read with exception {ve!r}; it is probably empty.")
    audio = data.T if data.ndim == 2 else data            # (channels, samples) like librosa mono=False
    if audio.ndim == 2:
        audio = audio.mean(axis=0)                         # librosa.load defaults to mono=True
    if sr != JUKEBOX_SAMPLE_RATE:
        audio = resample(audio, int(sr), JUKEBOX_SAMPLE_RATE, res_type or os.environ.get("LLARK_RES_TYPE", "kaiser_best"))
    return _normalize(audio).astype(np.float32)


def maybe_pad_audio_to_max_len(audio: np.ndarray, expected: int = JUKEBOX_EXPECTED_SAMPLES_LEN) -> np.ndarray:
    if len(audio) < expected:
        audio = np.pad(audio, (0, expected - len(audio)))
    return audio


def get_z(audio, vqvae: VQVAE):
    expected = vqvae.sample_length
    # don't compute unnecessary discrete encodings
    assert len(audio) >= expected, f"expected samples with shape {expected}; got shape {audio.shape}."
    audio = audio[:expected]

    x = torch.from_numpy(np.ascontiguousarray(audio[np.newaxis, :, np.newaxis], dtype=np.float32)).to(vqvae.device)
    zs = vqvae.encode(x)

    z = zs[-1].flatten()[np.newaxis, :]

    if z.shape[-1] < vqvae.hps.n_ctx:
        raise ValueError("Audio file is not long enough")

    return z


def get_cond(hps, top_prior: TopPrior):
    """Constant conditioning of the reference (jukebox/main.py:71-98): 62 s total length rounded
    down to a token multiple, offset 0, artist/genre "unknown".  Returns sample 0 only, shaped
    (1, n_ctx, width) and (1, 1, width)."""
    r2t = top_prior.raw_to_tokens
    total = (int(62 * hps.sr) // r2t) * r2t
    hps.sample_length_cond = total
    meta = dict(artist="unknown", genre="unknown", total_length=total, offset=0, lyrics="lyrics go here!!!")
    labels = top_prior.labeller.get_batch_labels([meta for _ in range(hps.n_samples)], "cpu")
    x_cond, y_cond, _prime = top_prior.get_cond(None, top_prior.get_y(labels, 0))
    return x_cond[0, : top_prior.n_ctx][None, ...], y_cond[0][None, ...]


def get_final_activations(z, x_cond, y_cond, top_prior: TopPrior):
    x = z[:, : top_prior.n_ctx]

    # make sure that we get the activations
    top_prior.prior.only_encode = True

    # encoder_kv and fp16 are set to the defaults, but explicitly so
    out = top_prior.prior.forward(x, x_cond=x_cond, y_cond=y_cond, encoder_kv=None, fp16=False)

    return out


def windowed_average(acts, frame_len: int, ceil_mode=False):
    """Windowed average over time: [T, W] -> [1, T // frame_len, W] (AvgPool1d semantics)."""
    assert acts.ndim == 2, "expected 2d inputs"
    if ceil_mode:
        raise NotImplementedError("ceil_mode=True is never used by the reference path (jukebox/main.py:113)")
    frames = acts.shape[0] // frame_len
    return ops.pool_window(acts.contiguous()[None], frame_len, frames)


def _postprocess(acts: torch.Tensor, latent_audio_len: int, meanpool: bool, pool_frames_per_second,
                 acts_sample_rate: float) -> np.ndarray:
    """jukebox/main.py:147-169 for one clip; acts: (n_ctx, width) on device."""
    acts = acts[:latent_audio_len, :]
    if meanpool:
        logging.warning(f"mean pooling at f={pool_frames_per_second}")
        if not pool_frames_per_second:
            acts = ops.pool_mean(acts.contiguous()[None])[0]
        else:
            frame_len = floor(acts_sample_rate / pool_frames_per_second)
            acts = windowed_average(acts, frame_len)
            acts = torch.squeeze(acts, 0)
    acts = acts.cpu().numpy().copy()
    logging.info(f"acts after pooling has shape {acts.shape}")
    return acts


def get_acts_from_audio(audio: np.ndarray, hps, vqvae, top_prior, meanpool=True, pool_frames_per_second=None):
    """``get_acts_from_file`` after the decode step (waveform already mono / normalised)."""
    expected = vqvae.sample_length
    n_ctx = top_prior.n_ctx
    input_audio_len = len(audio)
    latent_audio_len = floor(n_ctx * input_audio_len / expected)
    audio = maybe_pad_audio_to_max_len(audio, expected)

    z = get_z(audio, vqvae)                       # [1, T]
    x_cond, y_cond = get_cond(hps, top_prior)
    acts = get_final_activations(z, x_cond, y_cond, top_prior)
    acts = acts.squeeze(0).type(torch.float32)
    acts_rate = n_ctx / (expected / hps.sr)
    return _postprocess(acts, min(latent_audio_len, n_ctx), meanpool, pool_frames_per_second, acts_rate)


def get_acts_from_file(fpath, hps, vqvae, top_prior, meanpool=True, pool_frames_per_second=None):
    audio = load_audio_from_file(fpath)
    return get_acts_from_audio(audio, hps, vqvae, top_prior, meanpool, pool_frames_per_second)


def get_acts_from_audio_batch(audios: Sequence[np.ndarray], hps, vqvae, top_prior, meanpool=True,
                              pool_frames_per_second=None):
    """Batched ``get_acts_from_file`` (NEW: the reference processes one file per iteration).
    Every clip goes through exactly the per-clip arithmetic of :func:`get_acts_from_audio`; clips
    never interact, so results equal the one-at-a-time path."""
    expected = vqvae.sample_length
    n_ctx = top_prior.n_ctx
    lens, rows = [], []
    for a in audios:
        lens.append(min(floor(n_ctx * len(a) / expected), n_ctx))
        a = maybe_pad_audio_to_max_len(a, expected)
        assert len(a) >= expected
        rows.append(np.ascontiguousarray(a[:expected], dtype=np.float32))
    x = torch.from_numpy(np.stack(rows)).to(vqvae.device)
    z = vqvae.encode(x[:, :, None])[-1]
    x_cond, y_cond = get_cond(hps, top_prior)
    acts = get_final_activations(z, x_cond, y_cond, top_prior)
    acts_rate = n_ctx / (expected / hps.sr)
    return [_postprocess(acts[i], lens[i], meanpool, pool_frames_per_second, acts_rate) for i in range(len(audios))]


def load_model(model="5b", weights=None, hps: Optional[JukeboxHParams] = None, device="cuda", depth=None,
               restore_vqvae: Optional[str] = None, restore_prior: Optional[str] = None, precision: Optional[str] = None):
    """Builds (hps, vqvae, top_prior) like jukebox/main.py:176-200.

    ``precision`` (NEW): how the prior's Conv1D products carry the fp32 activation -- "f16x2" or "lo8", see
    ``llark_amd.jukebox.prior.PriorTransformer``; default ``$LLARK_PRIOR_PRECISION`` / the library default.

    ``weights=None`` (the reference's call, ``load_model()``): read upstream's ``5b/vqvae.pth.tar`` and
    ``<model>/prior_level_2.pth.tar`` from the local mirror ``~/.cache/jukebox/models`` (``$JUKEBOX_CACHE``) or from
    ``restore_vqvae`` / ``restore_prior``, with the reference's ``strict=False`` semantics: the checkpoint's prior layers
    >= 36 are dropped (jukebox/main.py:196, make_models.py.patch:7-8) -- see ``llark_amd/jukebox/checkpoint.py``.  A
    missing file raises ``FileNotFoundError`` (the reference would download it).
    ``weights`` = a state-dict-like mapping with upstream key names: used as given (extra keys ignored the same way).
    ``weights="synthetic"``: seeded synthetic weights (``llark_amd.jukebox.synthetic``; bench / tests, no checkpoint).
    ``setup_dist_from_mpi`` of the reference initialises a world-size-1 NCCL group that no collective ever uses
    (SURVEY 2b); nothing is initialised here.
    """
    if model not in ("5b", "5b_lyrics"):
        raise ValueError(f"unknown model {model!r}")
    hps = hparams_5b() if hps is None else hps
    hps.n_samples = 3 if model == "5b_lyrics" else 8
    d = hps.prior_depth if depth is None else depth
    if isinstance(weights, str):
        if weights != "synthetic":
            raise ValueError(f"weights must be a mapping, None or 'synthetic', got {weights!r}")
        from .synthetic import make_jukebox_weights

        weights = make_jukebox_weights(hps, seed=0, depth=depth)
    elif weights is None:
        from .checkpoint import load_checkpoint_weights

        weights, _unexpected = load_checkpoint_weights(model, hps, d, restore_vqvae, restore_prior)
    else:
        from .checkpoint import select_weights

        weights, _unexpected = select_weights([weights], hps, d, origin="weights mapping")
    vqvae = VQVAE(hps, weights, device)
    top_prior = TopPrior(hps, weights, device, depth=depth, precision=precision)
    return hps, vqvae, top_prior


class WrappedAudioEncoder(torch.nn.Module):
    """NEW (named by BASELINE.json): ``forward(audio (B, L) fp32) -> (B, frames, width)`` fp32 on device,
    wrapping get_z -> get_cond -> get_final_activations -> windowed pooling (10 fps => 240 frames for
    >= 23.8 s clips).  Clips are independent: this is the unit that shards across GPUs."""

    def __init__(self, hps=None, weights=None, device="cuda", pool_frames_per_second: int = 10, depth=None,
                 precision: Optional[str] = None):
        super().__init__()
        self.hps, self.vqvae, self.top_prior = load_model("5b", weights, hps, device, depth, precision=precision)
        self.pool_frames_per_second = pool_frames_per_second
        self._cond = None

    @property
    def frame_len(self) -> int:
        hps = self.hps
        acts_rate = hps.n_ctx / (hps.sample_length / hps.sr)
        return floor(acts_rate / self.pool_frames_per_second)

    @torch.no_grad()
    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        hps = self.hps
        assert audio.dim() == 2 and audio.shape[1] >= hps.sample_length, (
            f"expected (B, >= {hps.sample_length}) audio, got {tuple(audio.shape)}")
        x = audio[:, : hps.sample_length].to(device=self.vqvae.device, dtype=torch.float32).contiguous()
        z = self.vqvae.encode_top(x)
        if self._cond is None:
            self._cond = get_cond(hps, self.top_prior)
        acts = get_final_activations(z, self._cond[0], self._cond[1], self.top_prior)
        fl = self.frame_len
        return ops.pool_window(acts, fl, hps.n_ctx // fl)


def _select_shard(paths, batch_size, batch_idx):
    """--batch_size/--batch_idx shard the sorted file list (jukebox/main.py:227-232)."""
    if batch_size is None or batch_idx is None:
        return paths
    starts = range(0, len(paths), batch_size)
    if batch_idx >= len(starts):
        raise ValueError("Invalid batch index")
    return paths[starts[batch_idx] : starts[batch_idx] + batch_size]


def main(argv=None):
    """CLI with the reference's flags (jukebox/main.py:203-254): one ``<name>.npy`` per input wav."""
    import argparse
    import pathlib

    ap = argparse.ArgumentParser(description="Jukebox embeddings on MI355X")
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--batch_idx", type=int, default=None)
    ap.add_argument("--input_dir", default="/input", help="path to inputs")
    ap.add_argument("--output_dir", default="/output", help="path to outputs")
    ap.add_argument("--pool-frames-per-second", default=10, type=int,
                    help="Frames per second for pooling. Set to zero to pool over all timesteps.")
    ap.add_argument("--synthetic-weights", action="store_true",
                    help="NEW: seeded synthetic weights instead of the 5b checkpoint files (there are none offline)")
    ap.add_argument("--model", default="5b", choices=["5b", "5b_lyrics"], help="NEW: the reference hard-codes load_model()'s default")
    args = ap.parse_args(argv)

    out_dir = pathlib.Path(args.output_dir)
    out_dir.mkdir(exist_ok=True)
    paths = _select_shard(sorted(pathlib.Path(args.input_dir).iterdir()), args.batch_size, args.batch_idx)
    model = None
    for path in paths:
        if model is None:
            model = load_model(args.model, weights="synthetic" if args.synthetic_weights else None)
        hps, vqvae, top_prior = model
        with torch.no_grad():
            rep = get_acts_from_file(path, hps, vqvae, top_prior, meanpool=True,
                                     pool_frames_per_second=args.pool_frames_per_second)
        np.save(os.path.join(out_dir, os.path.basename(path).replace(".wav", ".npy")), rep)


if __name__ == "__main__":
    main()
