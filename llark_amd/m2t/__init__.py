"""LLM half of the hot path: drop-in mirrors of the reference's ``m2t/models`` on the HIP engine."""
from dataclasses import dataclass

from .special_tokens import DEFAULT_AUDIO_END_TOKEN, DEFAULT_AUDIO_PATCH_TOKEN, DEFAULT_AUDIO_START_TOKEN


@dataclass
class AudioEncoderConfig:
    """Token-id holder of the audio plugin surface (m2t/models/__init__.py:23-29).  Fields start as the
    token STRINGS and are overwritten with ids by ``initialize_audio_tokenizer`` like in the reference."""

    use_audio_start_end: bool = True
    audio_start_token: object = DEFAULT_AUDIO_START_TOKEN
    audio_end_token: object = DEFAULT_AUDIO_END_TOKEN
    audio_patch_token: object = DEFAULT_AUDIO_PATCH_TOKEN
