"""The model-side half of the reference's Beam pipeline (``jukebox/dataflow_inference.py``) on the MI355X path.

Mirrored (same names, arguments, return values and error behaviour):

  JukeboxModelWrapper(model_name, device)            jukebox/dataflow_inference.py:73-115
      .__call__(input_path) -> np.ndarray | None      (EmptyFileError -> None, :113-115; meanpool at 10 frames/s, :104-110)
  JukeboxModelHandler.load_model / run_inference      :118-158   (the Beam ``ModelHandler`` protocol, without Beam)
  get_input_file_list, write_prediction_result        :161-205   (local paths)
  read_wav_bytes                                       :66-70

Out of scope ("out of scope"): the Beam runner / Dataflow options and the GCS client -- ``gs://`` paths
raise ``NotImplementedError`` instead of importing ``google.cloud.storage``.  The reference's worker decodes the file it
downloaded from GCS from memory (``io.BytesIO(wav_bytes)``); the same happens here for local files, so the decode path
under test is the one the pipeline uses.  ``run_files`` is the single-process driver standing in for the pipeline graph
(list -> RunInference -> write), with the file list sharded over ranks like ``extract.main``'s ``--batch_idx``.
"""
from __future__ import annotations

import io
import logging
import os
import pathlib
from collections import namedtuple
from typing import Any, Dict, Iterable, Optional, Sequence

import numpy as np
import torch

from . import extract as E
from .audio_decode import UnsupportedContainerError

# apache_beam.ml.inference.base.PredictionResult is a (example, inference) named tuple
PredictionResult = namedtuple("PredictionResult", ["example", "inference"])


def read_wav_bytes(filepath) -> bytes:
    """The bytes of a wav file (the reference reads them from GCS, :66-70)."""
    filepath = str(filepath)
    if filepath.startswith("gs://"):
        raise NotImplementedError("GCS I/O is outside the hot path (SURVEY section 8); pass a local path")
    with open(filepath, "rb") as f:
        return f.read()


class JukeboxModelWrapper:
    """Wrapper for a Jukebox embedding model (jukebox/dataflow_inference.py:73-115)."""

    def __init__(self, model_name: str, device, weights=None, hps=None, depth=None, precision=None) -> None:
        self.model = model_name
        self.device = device
        self.chunk_size = 16 if self.model == "5b_lyrics" else 32
        self.max_batch_size = 3 if self.model == "5b_lyrics" else 16
        # hps.sr / n_samples / levels / hop_fraction, make_vqvae(sample_length=1048576), prior_depth = 36 and make_prior
        # (:80-95) are what extract.load_model builds; ``weights`` / ``hps`` / ``depth`` exist for tests (no checkpoint offline)
        self.hps, self.vqvae, self.top_prior = E.load_model(model_name, weights=weights, hps=hps, device=device, depth=depth,
                                                            precision=precision)
        self.hparams = {"prior_depth": self.top_prior.prior.depth}

    def __call__(self, input_path: str, *args: Any, **kwds: Any) -> Optional[np.ndarray]:
        print(f"processing file {input_path} on device {self.device}")
        wav_bytes = read_wav_bytes(input_path)
        with torch.no_grad():
            try:
                representation = E.get_acts_from_file(io.BytesIO(wav_bytes), self.hps, self.vqvae, self.top_prior,
                                                      meanpool=True, pool_frames_per_second=10)
                return representation
            except E.EmptyFileError:
                return None
            except UnsupportedContainerError as e:
                # mp3 / ogg are legitimate inputs of the reference (librosa -> audioread) that this decoder does not read: one such file
                # must not abort the whole extraction run -- skipped with its own message, like an empty file
                logging.warning(f"{input_path}: unsupported container, no representation written ({e})")
                return None


class JukeboxModelHandler:
    """``ModelHandler[str, PredictionResult, JukeboxModelWrapper]`` (:118-158) without the Beam base class."""

    def __init__(self, model_name: str = "5b", **wrapper_kwargs):
        self.model_name = model_name
        self._wrapper_kwargs = wrapper_kwargs

    def load_model(self) -> JukeboxModelWrapper:
        """Loads and initializes a model for processing (the reference asserts a CUDA device, :131-135)."""
        if not torch.cuda.is_available():
            raise RuntimeError("JukeboxModelHandler.load_model: no GPU (the MI355X path has no CPU fallback)")
        device = torch.device("cuda", torch.cuda.current_device())
        print(f"device count is {torch.cuda.device_count()}")
        print(f"device name: {torch.cuda.get_device_name(device)}")
        return JukeboxModelWrapper(self.model_name, device, **self._wrapper_kwargs)

    def run_inference(self, batch: Sequence[str], model: JukeboxModelWrapper,
                      inference_args: Optional[Dict[str, Any]] = None) -> Iterable[PredictionResult]:
        predictions = []
        for filepath in batch:
            outputs = model(filepath)
            predictions.append([outputs])
        return [PredictionResult(x, y) for x, y in zip(batch, predictions)]


def get_input_file_list(input_dir: str, extension: str = ".wav") -> Sequence[str]:
    if str(input_dir).startswith("gs://"):
        raise NotImplementedError("GCS I/O is outside the hot path (SURVEY section 8); pass a local directory")
    input_paths = sorted(str(p) for p in pathlib.Path(input_dir).iterdir())
    return [x for x in input_paths if x.endswith(extension)]


def write_prediction_result(prediction_result: PredictionResult, output_dir) -> None:
    """Write a prediction result to a local path as a numpy array (``<name>.wav`` -> ``<name>.npy``, :171-205).
    The inference is the one-element list ``run_inference`` wraps around the model output; an empty file's ``None`` is
    written as such by the reference (np.save of ``[None]``) -- here it is skipped with a warning."""
    input_path = prediction_result.example
    representation = prediction_result.inference
    if str(output_dir).startswith("gs://"):
        raise NotImplementedError("GCS I/O is outside the hot path (SURVEY section 8); pass a local directory")
    if isinstance(representation, list) and len(representation) == 1 and representation[0] is None:
        logging.warning(f"{input_path}: empty file, no representation written")
        return
    output_filename = os.path.basename(str(input_path)).replace(".wav", ".npy")
    os.makedirs(output_dir, exist_ok=True)
    np.save(os.path.join(str(output_dir), output_filename), representation)


def run_files(input_dir: str, output_dir: str, handler: Optional[JukeboxModelHandler] = None, rank: int = 0,
              world_size: int = 1, batch_size: int = 16) -> int:
    """list -> RunInference -> write, in one process; rank r takes every world_size-th file (clips are independent:
    no collective).  Returns the number of representations written."""
    handler = handler or JukeboxModelHandler()
    files = list(get_input_file_list(input_dir))[rank::world_size]
    if not files:
        return 0
    model = handler.load_model()
    written = 0
    for i in range(0, len(files), batch_size):
        for res in handler.run_inference(files[i:i + batch_size], model):
            write_prediction_result(res, output_dir)
            written += res.inference[0] is not None
    return written
