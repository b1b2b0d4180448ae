"""``infer_with_prompt`` with the signature and behaviour of the reference's ``m2t/infer.py:99-152``:
prompt text + audio encoding -> conversation -> tokens (prompt part only) -> ``model.generate`` with the
"###" stopping criterion."""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence

import torch

from .generate import KeywordsStoppingCriteria
from .prompting import (DEFAULT_CONVERSATION_HEADER, concat_audio_token_and_prompt, extract_prompt_tokens,
                        preprocess_for_lm_mappable, preprocess_multimodal_mappable)


def infer_with_prompt(prompt_text: str, model, audio_encoding, end_seq: Sequence[int], multimodal_cfg: Dict[str, Any],
                      tokenizer, example_id: Optional[str] = None, audio_first: bool = False,
                      header: str = DEFAULT_CONVERSATION_HEADER, **generation_kwargs):
    prompt_text = concat_audio_token_and_prompt(prompt_text, audio_first)
    elem = {
        "audio_encoding": audio_encoding,
        "audio_encoding_shape": list(audio_encoding.shape),
        "example_id": example_id,
        "id": example_id,
        "conversations": [{"from": "human", "value": prompt_text}, {"from": "gpt", "value": "<empty>"}],
    }
    elem = preprocess_for_lm_mappable(preprocess_multimodal_mappable(elem, multimodal_cfg), tokenizer=tokenizer, header=header)
    enc = elem.pop("audio_encoding")
    if enc.dim() < 3:
        enc = enc.unsqueeze(0)
    input_ids = extract_prompt_tokens(elem.pop("input_ids"), end_seq)
    if input_ids.dim() < 2:
        input_ids = input_ids.unsqueeze(0)
    stop = KeywordsStoppingCriteria(keywords=["###"], tokenizer=tokenizer, input_ids=input_ids)
    elem.pop("labels", None)
    elem.pop("example_id", None)
    return model.generate(**generation_kwargs, input_ids=input_ids.cuda(), audio_encodings=enc.cuda(), stopping_criteria=[stop])
