"""``KeywordsStoppingCriteria`` with the behaviour of the reference's ``m2t/generate.py:18-44``: stop when the
last generated token is a single-token keyword id, or when the decoded continuation contains a keyword."""
from __future__ import annotations

import torch


class KeywordsStoppingCriteria:
    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = list(keywords)
        self.tokenizer = tokenizer
        self.input_ids = input_ids
        self.start_len = None
        self.keyword_ids = []
        for kw in self.keywords:
            ids = tokenizer(kw).input_ids
            if isinstance(ids, list) and len(ids) == 1:        # only keywords that are exactly one token
                self.keyword_ids.append(ids[0])

    def __call__(self, output_ids: torch.LongTensor, scores=None, **kwargs) -> bool:
        if self.start_len is None:
            self.start_len = self.input_ids.shape[1]
        last = output_ids[0, -1]
        if any(last == k for k in self.keyword_ids):
            return True
        text = self.tokenizer.batch_decode(output_ids[:, self.start_len:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)
