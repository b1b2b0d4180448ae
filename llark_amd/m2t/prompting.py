"""Host-side prompt / batch glue of the LLM half (tokenizer-agnostic), with the semantics of the reference's
``m2t/data_modules.py`` (``concat_audio_token_and_prompt`` :287-292, ``preprocess_multimodal_mappable``
:234-258, ``sentences_to_formatted_conversation`` :92-109, ``_tokenize_fn`` :57-78, ``_mask_targets`` :81-89,
``preprocess_for_lm_mappable`` :261-284, ``preprocess_encodings`` :180-186,
``DataCollatorForSupervisedDataset`` :189-222), ``m2t/conversation_utils.py:36-55`` and the system header of
``m2t/llava/conversation.py:237-271`` (``conv_v1_2``).  Integer outputs are pinned against the reference's own
functions in tests/golden/prompt_glue.json.  The webdataset / GCS readers that feed these functions in the
reference are out of scope (I/O machinery)."""
from __future__ import annotations

import copy
import logging
from dataclasses import dataclass
from typing import Any, Dict, List, Sequence

import torch

from .special_tokens import (DEFAULT_AUDIO_END_TOKEN, DEFAULT_AUDIO_PATCH_TOKEN, DEFAULT_AUDIO_START_TOKEN,
                             DEFAULT_AUDIO_TOKEN, IGNORE_INDEX)

SYSTEM_PROMPT = ("A chat between a curious human and an artificial intelligence assistant. "
                 "The assistant gives helpful, detailed, and polite answers to the human's questions.")
ROLES = ("Human", "Assistant")
DEFAULT_CONVERSATION_HEADER = f"{SYSTEM_PROMPT}\n\n"
_TURN_OPEN, _TURN_CLOSE = "### ", "\n"


def concat_audio_token_and_prompt(prompt: str, audio_first: bool) -> str:
    return f"{DEFAULT_AUDIO_TOKEN}\n{prompt}" if audio_first else f"{prompt}\n{DEFAULT_AUDIO_TOKEN}"


def audio_token_len(enc_shape: Sequence[int]) -> int:
    return enc_shape[1] if (len(enc_shape) == 3 and enc_shape[0] == 1) else enc_shape[0]


def preprocess_multimodal_mappable(e: Dict[str, Any], multimodal_cfg: Dict[str, Any]) -> Dict[str, Any]:
    """<audio> -> <audio_start> + <audio_patch> x frames + <audio_end> in every turn."""
    assert not multimodal_cfg["sep_audio_conv_front"], "sep_audio_conv_front is not implemented (nor is it in the reference)"
    expansion = DEFAULT_AUDIO_PATCH_TOKEN * audio_token_len(e["audio_encoding_shape"])
    if multimodal_cfg["use_audio_start_end"]:
        expansion = DEFAULT_AUDIO_START_TOKEN + expansion + DEFAULT_AUDIO_END_TOKEN
    e["conversations"] = [{**turn, "value": turn["value"].replace(DEFAULT_AUDIO_TOKEN, expansion)} for turn in e["conversations"]]
    return e


def sentences_to_formatted_conversation(header: str, source: List[Dict[str, str]], get_conversation: bool = True) -> str:
    """header + "### Human: ...\\n### Assistant: ...\\n" + "### ".  Rewrites each turn's value in place, like the
    reference (the per-turn token counts used for masking are taken from the rewritten strings)."""
    text = header
    for turn in source:
        who = {"human": ROLES[0], "gpt": ROLES[1]}.get(turn["from"].lower(), "unknown")
        turn["value"] = f"{_TURN_OPEN}{who}: {turn['value']}{_TURN_CLOSE}"
        if get_conversation:
            text += turn["value"]
    return text + _TURN_OPEN


def tokenize_strings(strings: Sequence[str], tokenizer) -> Dict[str, list]:
    toks = [tokenizer(s, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length, truncation=True)
            for s in strings]
    ids = [t.input_ids[0] for t in toks]
    lens = [int(t.input_ids.ne(tokenizer.pad_token_id).sum().item()) for t in toks]
    return dict(input_ids=ids, labels=ids, input_ids_lens=lens, labels_lens=lens)


def mask_targets(target: torch.Tensor, tokenized_lens: Sequence[int], speakers: Sequence[str]) -> None:
    """-100 over the header and over every human turn (keeping its first two tokens, as the reference does)."""
    pos = tokenized_lens[0]
    target[:pos] = IGNORE_INDEX
    for n, speaker in zip(tokenized_lens[1:], speakers):
        if speaker == "human":
            target[pos + 2: pos + n] = IGNORE_INDEX
        pos += n


def preprocess_encodings(audio_encoding, audio_encoding_shape: List[int]) -> torch.Tensor:
    enc = torch.Tensor(audio_encoding).reshape(audio_encoding_shape)
    return torch.squeeze(enc, 0) if enc.ndim == 3 else enc


def preprocess_for_lm_mappable(e: Dict[str, Any], tokenizer, header: str = DEFAULT_CONVERSATION_HEADER) -> Dict[str, Any]:
    source = e["conversations"]
    conversation = sentences_to_formatted_conversation(header, source)
    input_ids = tokenize_strings([conversation], tokenizer)["input_ids"][0]
    target = copy.deepcopy(input_ids)
    lens = tokenize_strings([header] + [t["value"] for t in source], tokenizer)["input_ids_lens"]
    mask_targets(target, lens, [t["from"] for t in source])
    return dict(input_ids=input_ids, labels=target,
                audio_encoding=preprocess_encodings(e["audio_encoding"], e["audio_encoding_shape"]), example_id=e["id"])


@dataclass
class DataCollatorForSupervisedDataset:
    """Right-pads input_ids with the pad id and labels with -100; stacks equal-shape encodings, else passes a list."""

    tokenizer: Any

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, Any]:
        pad = torch.nn.utils.rnn.pad_sequence
        input_ids = pad([i["input_ids"] for i in instances], batch_first=True, padding_value=self.tokenizer.pad_token_id)
        labels = pad([i["labels"] for i in instances], batch_first=True, padding_value=IGNORE_INDEX)
        batch = dict(input_ids=input_ids, labels=labels, attention_mask=input_ids.ne(self.tokenizer.pad_token_id))
        if "audio_encoding" in instances[0]:
            encs = [i["audio_encoding"] for i in instances]
            same = all(x is not None and x.shape == encs[0].shape for x in encs)
            batch["audio_encodings"] = torch.stack(encs) if same else encs
        else:
            logging.warning("key `audio_encoding` not detected in data collator inputs.")
        return batch


def _find_subsequence(seq: list, sub: list):
    n = len(sub)
    for start in range(len(seq) - n):
        if seq[start: start + n] == sub:
            return start, start + n
    return None


def extract_prompt_tokens(input_ids: torch.Tensor, end_seq: Sequence[int]) -> torch.Tensor:
    """input_ids up to and including the first occurrence of ``end_seq`` ("\\n### Assistant:")."""
    _, end = _find_subsequence(input_ids.tolist(), list(end_seq))
    return input_ids[:end]


def extract_response_tokens(input_ids: torch.Tensor, end_seq: Sequence[int]) -> torch.Tensor:
    _, end = _find_subsequence(input_ids.tolist(), list(end_seq))
    return input_ids[end:]
