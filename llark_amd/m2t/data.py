"""Training data path of the native loop: reads the reference's shard format without the ``webdataset`` dependency.

The reference streams tar shards (``m2t/data_modules.py:466-520`` ``read_webdataset``): every sample is a group of tar
members sharing a key -- ``<key>.json`` (``{"response": [{"question": ..., "answer": ...}, ...]}``) and
``<key>.audio_encoding.pyd`` (a pickled array) -- which ``webdataset_element_to_conversation`` (:295-340) unpacks into one
conversation per question/answer pair with the ``<audio>`` placeholder randomly first or last; the conversations then go
through ``preprocess_multimodal_mappable`` / ``preprocess_for_lm_mappable`` and the collator (llark_amd.m2t.prompting).

Here: a pure-``tarfile`` reader for local shards (``.npy`` members are read natively; ``.pyd`` pickles only with
``allow_pickle=True`` -- unpickling is code execution, the caller must trust the shards), brace expansion of
``name-{000..127}.tar`` lists (:436-438), the per-rank shard split of ``wds.split_by_node``, and a micro-batch generator
for ``llark_amd.m2t.train.train``; with ``task_sample_probs`` the shard order of an epoch is drawn WITH replacement, each shard
weighted by the probability of the task whose name its path contains (``repeat_shards``, :441-463; off unless
``--apply_task_sample_probs``, as in m2t/arguments.py:61-68).  GCS URLs and the HF ``IterableDataset`` wrapper are not built
(I/O glue outside the hot path).
"""
from __future__ import annotations

import io
import json
import pickle
import random
import re
import tarfile
from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .prompting import (DataCollatorForSupervisedDataset, concat_audio_token_and_prompt, preprocess_for_lm_mappable,
                        preprocess_multimodal_mappable)

_RANGE = re.compile(r"\{(\d+)\.\.(\d+)\}")


def expand_urls(url: str) -> List[str]:
    """``"a-{000..002}.tar,b.tar"`` -> ``[a-000.tar, a-001.tar, a-002.tar, b.tar]`` (numeric ranges, zero padding kept)."""
    out: List[str] = []
    for part in url.split(","):
        part = part.strip()
        if not part:
            continue
        todo = [part]
        while todo:
            cur = todo.pop(0)
            m = _RANGE.search(cur)
            if not m:
                out.append(cur)
                continue
            lo, hi, width = int(m.group(1)), int(m.group(2)), len(m.group(1))
            todo = [cur[: m.start()] + str(i).zfill(width) + cur[m.end():] for i in range(lo, hi + 1)] + todo
    return out


DEFAULT_TASK_SAMPLE_PROBS = {"captioning": 0.15, "reasoning": 0.55, "mir": 0.3}      # m2t/arguments.py:61-67


def shard_probs(urls: Sequence[str], task_sample_probs: Dict[str, float]) -> List[float]:
    """``repeat_shards`` (m2t/data_modules.py:441-457): a shard's weight is the probability of the FIRST task key contained in
    its name, normalised over the list; a shard matching no key is an error."""
    probs = []
    for shard in urls:
        for k, prob in task_sample_probs.items():
            if k in shard:
                probs.append(float(prob))
                break
        else:
            raise ValueError(f"probability for shard {shard} not defined in probs {task_sample_probs}")
    total = sum(probs)
    if total <= 0:
        raise ValueError(f"task probabilities {task_sample_probs} give the shard list zero total weight")
    return [p / total for p in probs]


def split_by_rank(urls: Sequence[str], rank: int, world: int) -> List[str]:
    """``wds.split_by_node``: rank r reads shards r, r + world, ...  (every rank needs at least one shard)."""
    mine = list(urls[rank::world])
    if not mine:
        raise ValueError(f"{len(urls)} shard(s) cannot be split over {world} ranks")
    return mine


def _decode_member(name: str, data: bytes, allow_pickle: bool):
    if name.endswith(".json"):
        return json.loads(data.decode("utf-8"))
    if name.endswith(".npy"):
        return np.load(io.BytesIO(data), allow_pickle=False)
    if name.endswith(".pyd"):
        if not allow_pickle:
            raise ValueError(f"{name}: pickled member; pass allow_pickle=True only for shards you trust (or store .npy)")
        obj = pickle.loads(data)
        return obj.numpy() if isinstance(obj, torch.Tensor) else np.asarray(obj)
    return data


def iter_tar_samples(shards: Iterable[str], allow_pickle: bool = False) -> Iterator[Dict[str, Any]]:
    """Yields ``{"__key__", "__url__", "json", "audio_encoding.<ext>", ...}`` per key, in tar order (members of one sample
    are adjacent, as webdataset requires)."""
    for url in shards:
        with tarfile.open(url, "r:*") as tf:
            cur: Dict[str, Any] = {}
            for member in tf:
                if not member.isfile():
                    continue
                base = member.name.rsplit("/", 1)[-1]
                key, _, ext = base.partition(".")
                if cur and cur["__key__"] != key:
                    yield cur
                    cur = {}
                if not cur:
                    cur = {"__key__": key, "__url__": url}
                cur[ext] = _decode_member(base, tf.extractfile(member).read(), allow_pickle)
            if cur:
                yield cur


def element_to_conversations(elem: Dict[str, Any], rng: random.Random) -> Iterator[Dict[str, Any]]:
    """m2t/data_modules.py:295-340: one training example per (question, answer) of a sample; malformed samples are skipped."""
    js = elem.get("json")
    if not isinstance(js, dict) or not isinstance(js.get("response"), list) or not js["response"]:
        return
    enc = elem.get("audio_encoding.pyd", elem.get("audio_encoding.npy"))
    if enc is None or not hasattr(enc, "shape"):
        return
    for resp in js["response"]:
        audio_first = rng.uniform(0.0, 1.0) > 0.5
        yield {"audio_encoding": enc, "audio_encoding_shape": list(enc.shape), "id": elem["__key__"],
               "conversations": [{"from": "human", "value": concat_audio_token_and_prompt(resp["question"], audio_first)},
                                 {"from": "gpt", "value": resp["answer"]}]}


SHUFFLE_BUFFER = 100          # m2t/data_modules.py: dataset.shuffle(100) after webdataset_element_to_conversation


def _shuffled(items: Iterator[Dict[str, Any]], rng: random.Random, size: int) -> Iterator[Dict[str, Any]]:
    """webdataset's bounded shuffle: keep ``size`` items, emit a random one as each new item arrives, drain at the end."""
    buf: List[Dict[str, Any]] = []
    for it in items:
        if len(buf) < size:
            buf.append(it)
            continue
        j = rng.randrange(size)
        buf[j], it = it, buf[j]
        yield it
    rng.shuffle(buf)
    yield from buf


def micro_batches(train_data_path: str, tokenizer, multimodal_cfg: Dict[str, Any], batch_size: int, model_max_length: int,
                  rank: int = 0, world: int = 1, seed: int = 0, epochs: Optional[int] = None, allow_pickle: bool = False,
                  shuffle_buffer: int = SHUFFLE_BUFFER, skip_micro_batches: int = 0,
                  task_sample_probs: Optional[Dict[str, float]] = None):
    """Collated micro-batches (``input_ids``, ``labels``, ``attention_mask``, ``audio_encodings``) of THIS rank, forever
    (``epochs=None``, like the reference's ``repeat()``) or for a number of passes over its shards.

    Order (m2t/data_modules.py:560-640): shards shuffled per epoch, then the (question, answer) examples go through a
    bounded shuffle buffer of ``shuffle_buffer`` conversations so that the pairs of one clip are not emitted back to back;
    both generators are seeded per rank AND per epoch.  ``skip_micro_batches`` fast-forwards the stream (resume: the
    trainer passes ``step * gradient_accumulation_steps``) so a resumed run continues where the interrupted one stopped
    instead of replaying its first samples.  ``task_sample_probs``: see :func:`shard_probs` -- an epoch then reads as many shards
    as the rank owns, drawn with replacement by task weight over the WHOLE shard list (one draw shared by all ranks, then split
    r::world; the reference draws 1024 x len(urls) up front, before split_by_node)."""
    all_shards = expand_urls(train_data_path)
    shards = split_by_rank(all_shards, rank, world)
    # task weights are normalised over the GLOBAL shard list and the weighted order is drawn over it with a RANK-INDEPENDENT seed, then
    # split by rank -- the reference's repeat_shards() runs before split_by_node (m2t/data_modules.py:441-462, 560-640), so every rank
    # sees the task mix of the whole list, also when its own r::world subset is skewed or lacks a task
    weights = shard_probs(all_shards, task_sample_probs) if task_sample_probs else None
    collate = DataCollatorForSupervisedDataset(tokenizer)
    epoch = 0
    to_skip = max(0, int(skip_micro_batches))                         # whole micro-batches still to fast-forward over
    while epochs is None or epoch < epochs:
        rng = random.Random((seed * 1000003 + rank) * 7919 + epoch)
        if weights is None:
            order = list(shards)
            rng.shuffle(order)                                        # shardshuffle
        else:
            grng = random.Random(seed * 1000003 * 7919 + epoch)          # the same draw on every rank
            order = grng.choices(all_shards, weights=weights, k=len(shards) * world)[rank::world]

        def conversations():
            for elem in iter_tar_samples(order, allow_pickle):
                yield from element_to_conversations(elem, rng)

        stream = _shuffled(conversations(), rng, shuffle_buffer) if shuffle_buffer and shuffle_buffer > 1 else conversations()
        pending: List[Dict[str, Any]] = []
        skip_pending = 0      # conversations of the micro-batch being skipped; like `pending`, an epoch's remainder is dropped
        for conv in stream:
            if to_skip > 0:                                           # fast-forward without tokenising
                skip_pending += 1
                if skip_pending == batch_size:
                    to_skip -= 1
                    skip_pending = 0
                continue
            ex = preprocess_for_lm_mappable(preprocess_multimodal_mappable(conv, multimodal_cfg), tokenizer=tokenizer)
            ex["input_ids"], ex["labels"] = ex["input_ids"][:model_max_length], ex["labels"][:model_max_length]
            pending.append(ex)
            if len(pending) == batch_size:
                yield collate(pending)
                pending = []
        epoch += 1
