"""Drop-in for the object the reference's Beam handler drives (``scripts/clap/clap_embeddings.py:63-107``):
``laion_clap.CLAP_Module(enable_fusion=False, amodel="HTSAT-base")`` -> ``load_ckpt(path)`` ->
``model.model.get_audio_embedding([{"waveform": ...}, ...])`` -> ``(B, 512)`` L2-normalised embeddings.  Same names and
argument meaning; the compute is the HIP front end + HTSAT engine, and there is no CPU path."""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence

import numpy as np
import torch

from .frontend import CLIP_SAMPLES, ClapFrontend, fit_clip
from .htsat import ClapDims, HipClapAudioEncoder, from_laion_state_dict


class _AudioModel:
    """Stands where ``CLAP_Module.model`` (the CLAP nn.Module) stands: only the audio tower is provided."""

    def __init__(self, owner: "HipClapModule"):
        self._owner = owner

    def get_audio_embedding(self, data: Sequence[Dict[str, Any]]) -> torch.Tensor:
        own = self._owner
        if own.encoder is None:
            raise RuntimeError("no weights loaded: call load_ckpt() / load_state_dict() first")
        waves = []
        for d in data:
            w = d["waveform"]
            w = w.detach().to(torch.float32) if isinstance(w, torch.Tensor) else torch.as_tensor(np.asarray(w, np.float32))
            if w.numel() != CLIP_SAMPLES:
                raise ValueError(f"waveform must hold {CLIP_SAMPLES} samples (get_audio_features pads / crops to it), got {w.numel()}")
            waves.append(w.reshape(-1))
        wav = torch.stack(waves).to(own.device)
        # load_audio_input has already applied the int16 round trip when it built the dict: not repeated here
        return own.encoder.embed(own.frontend.logmel(wav, quantize_int16=False), normalize=True)


class HipClapModule:
    def __init__(self, enable_fusion: bool = False, amodel: str = "HTSAT-base", device="cuda:0", precision: str = "fp32",
                 dims: Optional[ClapDims] = None):
        if enable_fusion:
            raise NotImplementedError("enable_fusion=True (feature-fusion CLAP) is not used by the reference pipeline")
        if amodel != "HTSAT-base":
            raise NotImplementedError(f"amodel {amodel!r}: the reference pipeline uses HTSAT-base")
        self.device = torch.device(device)
        self.precision = precision
        self.dims = dims or ClapDims()
        self.frontend = ClapFrontend(self.device)
        self.encoder: Optional[HipClapAudioEncoder] = None
        self.model = _AudioModel(self)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Either laion_clap names (``[module.]audio_branch.*``) or the transformers names the engine uses."""
        if any(k.startswith(("module.audio_branch.", "audio_branch.")) for k in sd):
            sd = from_laion_state_dict(sd)
        self.encoder = HipClapAudioEncoder(sd, self.dims, self.device, self.precision)

    def load_ckpt(self, ckpt: str) -> None:
        blob = torch.load(ckpt, map_location="cpu", weights_only=True)
        self.load_state_dict(blob["state_dict"] if isinstance(blob, dict) and "state_dict" in blob else blob)

    def get_audio_embedding_from_data(self, x: Sequence[np.ndarray], use_tensor: bool = False, rng: Optional[np.random.Generator] = None):
        """laion_clap's convenience entry: raw 48 kHz clips of any length -> embeddings (int16 round trip, rand_trunc /
        repeatpad, then the model)."""
        if self.encoder is None:
            raise RuntimeError("no weights loaded: call load_ckpt() / load_state_dict() first")
        wav = self.frontend.batch(x, rng)
        out = self.encoder.embed(self.frontend.logmel(wav, quantize_int16=True), normalize=True)
        return out if use_tensor else out.cpu().numpy()


def load_audio_input(elem: Dict[str, Any], rng: Optional[np.random.Generator] = None) -> Dict[str, Any]:
    """``load_audio_input`` of the reference (clap_embeddings.py:127-153) for an element that already carries its 48 kHz
    samples as ``elem["waveform"]`` (file reading / resampling stays with the caller's loader): int16 round trip, then
    rand_trunc / repeatpad -> ``elem["audio_features"] = [{"waveform": tensor(480000)}]``."""
    w = np.clip(np.asarray(elem["waveform"], np.float32), -1.0, 1.0)
    w = ((w * 32767.0).astype(np.int16) / 32767.0).astype(np.float32)
    elem["audio_features"] = [{"waveform": torch.from_numpy(fit_clip(w, rng))}]
    return elem
