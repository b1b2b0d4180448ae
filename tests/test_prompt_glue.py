"""CPU: host-side prompt glue vs golden vectors produced by the reference's own functions
(tests/golden/prompt_glue.json <- tests/golden/make_prompt_golden.py). Integer work -> exact."""
import json
import os

import numpy as np
import torch

from llark_amd.m2t import prompting as P
from llark_amd.m2t.generate import KeywordsStoppingCriteria
from toy_tokenizer import ToyTokenizer

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prompt_glue.json")))


def _tok():
    tok = ToyTokenizer()
    tok.vocab = dict(GOLD["vocab"])
    tok.inv = {v: k for k, v in tok.vocab.items()}
    tok.special = ["<audio_patch>", "<audio_start>", "<audio_end>"]
    return tok


def test_header_and_concat():
    assert P.DEFAULT_CONVERSATION_HEADER == GOLD["header"]
    for c in GOLD["cases"]:
        assert P.concat_audio_token_and_prompt(c["prompt"], c["audio_first"]) == c["text"]


def test_tokens_labels_prompt_split():
    tok = _tok()
    cfg = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)
    for i, c in enumerate(GOLD["cases"]):
        enc = np.arange(c["frames"] * 4, dtype=np.float32).reshape(c["frames"], 4)
        elem = {"audio_encoding": enc, "audio_encoding_shape": list(enc.shape), "example_id": f"ex{i}", "id": f"ex{i}",
                "conversations": [{"from": "human", "value": c["text"]}, {"from": "gpt", "value": "a slow ballad" if i else "<empty>"}]}
        elem = P.preprocess_multimodal_mappable(elem, cfg)
        assert [dict(t) for t in elem["conversations"]] == c["conversations"]
        res = P.preprocess_for_lm_mappable(elem, tokenizer=tok)
        assert res["input_ids"].tolist() == c["input_ids"]
        assert res["labels"].tolist() == c["labels"]
        assert list(res["audio_encoding"].shape) == c["audio_shape"]
        assert P.extract_prompt_tokens(res["input_ids"], c["end_seq"]).tolist() == c["prompt_ids"]
        assert P.extract_response_tokens(res["input_ids"], c["end_seq"]).tolist() == c["response_ids"]
        # layout rule of the splice (m2t/models/llamav2.py:169-175): ids[start + F + 1] == <audio_end>
        ids = res["input_ids"].tolist()
        start = ids.index(tok.vocab["<audio_start>"])
        assert ids[start + c["frames"] + 1] == tok.vocab["<audio_end>"]


def test_collator():
    tok = _tok()
    cases = GOLD["cases"]
    inst = [dict(input_ids=torch.tensor(c["input_ids"]), labels=torch.tensor(c["labels"]),
                 audio_encoding=torch.zeros(c["frames"], 4)) for c in cases]
    b = P.DataCollatorForSupervisedDataset(tokenizer=tok)(inst)
    g = GOLD["collated"]
    assert b["input_ids"].tolist() == g["input_ids"] and b["labels"].tolist() == g["labels"]
    assert b["attention_mask"].long().tolist() == g["attention_mask"]
    assert isinstance(b["audio_encodings"], list) == g["encodings_is_list"]
    inst2 = [dict(input_ids=torch.tensor(cases[0]["input_ids"]), labels=torch.tensor(cases[0]["labels"]),
                  audio_encoding=torch.zeros(5, 4)) for _ in range(2)]
    b2 = P.DataCollatorForSupervisedDataset(tokenizer=tok)(inst2)
    assert torch.is_tensor(b2["audio_encodings"]) == GOLD["collated_equal_shapes_is_tensor"]
    # right padding only: the mask form the HIP engine accepts
    am = b["attention_mask"]
    assert bool((am[:, :-1] | ~am[:, 1:]).all())


def test_stopping_criterion():
    tok = _tok()
    ids0 = torch.tensor([GOLD["cases"][0]["prompt_ids"]])
    for name, seq in GOLD["stopping_seqs"].items():
        crit = KeywordsStoppingCriteria(keywords=["###"], tokenizer=tok, input_ids=ids0)
        assert bool(crit(torch.tensor([seq]), None)) == GOLD["stopping"][name]
