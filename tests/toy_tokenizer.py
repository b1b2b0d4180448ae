"""A tiny deterministic tokenizer (test infrastructure) that quacks like the slice of the HF tokenizer API the
reference's prompt glue uses: call -> .input_ids (pt), pad/bos ids, model_max_length, add_tokens,
convert_tokens_to_ids, decode / batch_decode.  Words, newlines and added special tokens are single tokens;
a BOS is prepended to every encoded string (like the Llama SentencePiece tokenizer)."""
import re
from types import SimpleNamespace

import torch


class ToyTokenizer:
    def __init__(self, model_max_length=2048):
        self.vocab = {"[PAD]": 0, "<s>": 1, "</s>": 2, "<unk>": 3}
        self.inv = {v: k for k, v in self.vocab.items()}
        self.special = []
        self.pad_token_id, self.bos_token_id, self.eos_token_id = 0, 1, 2
        self.model_max_length = model_max_length
        self.padding_side = "right"

    def __len__(self):
        return len(self.vocab)

    def _id(self, tok):
        if tok not in self.vocab:
            self.vocab[tok] = len(self.vocab)
            self.inv[self.vocab[tok]] = tok
        return self.vocab[tok]

    def add_tokens(self, toks, special_tokens=False):
        n = 0
        for t in toks:
            if t not in self.vocab:
                self._id(t)
                n += 1
            if t not in self.special:
                self.special.append(t)
        return n

    def convert_tokens_to_ids(self, toks):
        return [self.vocab[t] for t in toks]

    def _split(self, text):
        pat = "|".join(re.escape(s) for s in sorted(self.special, key=len, reverse=True))
        pat = (pat + "|" if pat else "") + r"\n|[^\s<]+|<"
        return re.findall(pat, text)

    def encode(self, text):
        return [self.bos_token_id] + [self._id(t) for t in self._split(text)]

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=False, **kw):
        ids = self.encode(text)
        if truncation and max_length:
            ids = ids[:max_length]
        if return_tensors == "pt":
            return SimpleNamespace(input_ids=torch.tensor([ids], dtype=torch.long))
        return SimpleNamespace(input_ids=ids)

    def decode(self, ids, skip_special_tokens=False):
        out = []
        for i in (ids.tolist() if hasattr(ids, "tolist") else ids):
            t = self.inv[int(i)]
            if skip_special_tokens and (t in self.special or int(i) < 4):
                continue
            out.append(t)
        return " ".join(out)

    def batch_decode(self, batch, skip_special_tokens=False):
        return [self.decode(row, skip_special_tokens) for row in batch]
