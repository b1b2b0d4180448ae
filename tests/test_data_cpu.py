"""llark_amd.m2t.data: the reference's shard format (m2t/data_modules.py:295-340,436-520) read without webdataset."""
import io
import json
import os
import pickle
import random
import sys
import tarfile

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from toy_tokenizer import ToyTokenizer  # noqa: E402

from llark_amd.m2t import data as D  # noqa: E402


def _add(tf, name, payload: bytes):
    info = tarfile.TarInfo(name)
    info.size = len(payload)
    tf.addfile(info, io.BytesIO(payload))


def _npy(a):
    b = io.BytesIO()
    np.save(b, a)
    return b.getvalue()


def _make_shard(path, keys, frames=3, pyd=False, rng=None):
    rng = rng or np.random.default_rng(0)
    with tarfile.open(path, "w") as tf:
        for k in keys:
            enc = rng.standard_normal((frames, 8)).astype(np.float32)
            resp = {"response": [{"question": f"what is {k} ?", "answer": f"it is {k} ."}, {"question": "tempo ?", "answer": "fast ."}]}
            _add(tf, f"{k}.json", json.dumps(resp).encode())
            if pyd:
                _add(tf, f"{k}.audio_encoding.pyd", pickle.dumps(torch.from_numpy(enc)))
            else:
                _add(tf, f"{k}.audio_encoding.npy", _npy(enc))


def test_expand_and_split():
    assert D.expand_urls("a-{000..002}.tar, b.tar") == ["a-000.tar", "a-001.tar", "a-002.tar", "b.tar"]
    assert D.expand_urls("x{8..10}-{0..1}.tar") == ["x8-0.tar", "x8-1.tar", "x9-0.tar", "x9-1.tar", "x10-0.tar", "x10-1.tar"]
    urls = [f"s{i}" for i in range(5)]
    assert D.split_by_rank(urls, 0, 2) == ["s0", "s2", "s4"] and D.split_by_rank(urls, 1, 2) == ["s1", "s3"]
    with pytest.raises(ValueError):
        D.split_by_rank(["s0"], 1, 2)


def test_tar_reader_conversations_and_batches(tmp_path):
    _make_shard(tmp_path / "d-000.tar", ["k0", "k1"])
    _make_shard(tmp_path / "d-001.tar", ["k2"], pyd=True)
    with tarfile.open(tmp_path / "d-002.tar", "w") as tf:                 # malformed samples are skipped, not fatal
        _add(tf, "bad0.json", json.dumps({"response": "oops"}).encode())
        _add(tf, "bad0.audio_encoding.npy", _npy(np.zeros((3, 8), np.float32)))
        _add(tf, "bad1.json", json.dumps({"response": [{"question": "q", "answer": "a"}]}).encode())      # no encoding
    elems = list(D.iter_tar_samples([str(tmp_path / "d-000.tar")]))
    assert [e["__key__"] for e in elems] == ["k0", "k1"] and elems[0]["audio_encoding.npy"].shape == (3, 8)
    with pytest.raises(ValueError, match="allow_pickle"):
        list(D.iter_tar_samples([str(tmp_path / "d-001.tar")]))
    e2 = list(D.iter_tar_samples([str(tmp_path / "d-001.tar")], allow_pickle=True))[0]
    assert isinstance(e2["audio_encoding.pyd"], np.ndarray)
    convs = list(D.element_to_conversations(elems[0], random.Random(0)))
    assert len(convs) == 2 and convs[0]["id"] == "k0" and convs[0]["conversations"][1]["value"] == "it is k0 ."
    assert all(("<audio>" in c["conversations"][0]["value"]) for c in convs)
    assert list(D.element_to_conversations({"__key__": "x", "json": {"response": "oops"}}, random.Random(0))) == []
    # collated micro-batches through the reference's prompt glue
    tok = ToyTokenizer()
    tok.add_tokens(["<audio_patch>", "<audio_start>", "<audio_end>"], special_tokens=True)
    mm = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)
    pattern = str(tmp_path / "d-{000..002}.tar")
    got = list(D.micro_batches(pattern, tok, mm, batch_size=2, model_max_length=64, epochs=1, allow_pickle=True))
    assert len(got) == 3                                                  # 3 good samples x 2 QA pairs = 6 examples
    b = got[0]
    assert set(b) == {"input_ids", "labels", "attention_mask", "audio_encodings"} and b["input_ids"].shape == b["labels"].shape
    patch = tok.convert_tokens_to_ids(["<audio_patch>"])[0]
    assert int((b["input_ids"][0] == patch).sum()) == 3                   # one patch token per frame
    assert (b["labels"] == -100).any() and (b["labels"] != -100).any()    # prompt masked, answer supervised
    assert b["input_ids"].shape[1] <= 64
    # two ranks see disjoint shards
    r0 = [x["input_ids"].shape[0] for x in D.micro_batches(pattern, tok, mm, 2, 64, rank=0, world=2, epochs=1, allow_pickle=True)]
    r1 = [x["input_ids"].shape[0] for x in D.micro_batches(pattern, tok, mm, 2, 64, rank=1, world=2, epochs=1, allow_pickle=True)]
    assert sum(r0) == 4 and sum(r1) == 2                                   # rank 0: d-000 (+ the malformed d-002), rank 1: d-001


def test_resume_fast_forward_matches_uninterrupted_stream_across_epochs(tmp_path):
    """an epoch whose conversation count is not a multiple of the batch size drops its remainder; the resume
    fast-forward must drop it too, or the resumed stream drifts by up to batch_size - 1 conversations per epoch."""
    _make_shard(tmp_path / "e-000.tar", ["a0", "a1", "a2"])                # 3 clips x 2 QA pairs = 6 conversations / epoch
    _make_shard(tmp_path / "e-001.tar", ["b0"])                            # + 2 = 8; batch 3 -> 2 batches + remainder 2
    tok = ToyTokenizer()
    tok.add_tokens(["<audio_patch>", "<audio_start>", "<audio_end>"], special_tokens=True)
    mm = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)
    pattern = str(tmp_path / "e-{000..001}.tar")
    full = list(D.micro_batches(pattern, tok, mm, batch_size=3, model_max_length=64, epochs=4, seed=5))
    assert len(full) == 8                                                  # 2 per epoch
    for skip in (1, 2, 3, 5):                                              # inside epoch 0, at its end, in epoch 1, in epoch 2
        rest = list(D.micro_batches(pattern, tok, mm, batch_size=3, model_max_length=64, epochs=4, seed=5, skip_micro_batches=skip))
        assert len(rest) == len(full) - skip
        for a, b in zip(rest, full[skip:]):
            assert torch.equal(a["input_ids"], b["input_ids"]) and torch.equal(a["labels"], b["labels"])


def test_task_sample_probs_weight_the_shard_draw(tmp_path):
    """``repeat_shards`` (m2t/data_modules.py:441-463, behind --apply_task_sample_probs): shards are drawn with replacement,
    weighted by the probability of the task named in their path; a shard of no known task is an error."""
    assert D.shard_probs(["a/captioning-000.tar", "b/mir-000.tar", "b/mir-001.tar"], {"captioning": 0.2, "mir": 0.4}) == [0.2, 0.4, 0.4]
    assert D.DEFAULT_TASK_SAMPLE_PROBS == {"captioning": 0.15, "reasoning": 0.55, "mir": 0.3}
    with pytest.raises(ValueError, match="not defined in probs"):
        D.shard_probs(["x/unknown-000.tar"], {"mir": 1.0})
    _make_shard(tmp_path / "mir-000.tar", ["m0"])                          # 1 clip x 2 QA pairs
    _make_shard(tmp_path / "captioning-000.tar", ["c0"])
    tok = ToyTokenizer()
    tok.add_tokens(["<audio_patch>", "<audio_start>", "<audio_end>"], special_tokens=True)
    mm = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)
    pattern = f"{tmp_path}/mir-000.tar,{tmp_path}/captioning-000.tar"
    only_mir = list(D.micro_batches(pattern, tok, mm, batch_size=2, model_max_length=64, epochs=6, seed=1,
                                    task_sample_probs={"mir": 1.0, "captioning": 0.0}))
    assert len(only_mir) == 12                                             # 2 draws per epoch, always the mir shard: 2 x 2 pairs
    plain = list(D.micro_batches(pattern, tok, mm, batch_size=2, model_max_length=64, epochs=6, seed=1))
    assert len(plain) == 12
    ids = lambda bs: sorted({tuple(r.tolist()) for b in bs for r in b["input_ids"]})
    assert len(ids(only_mir)) < len(ids(plain))                            # the captioning clip never appears
    a = list(D.micro_batches(pattern, tok, mm, 2, 64, epochs=3, seed=4, task_sample_probs={"mir": 0.5, "captioning": 0.5}))
    b = list(D.micro_batches(pattern, tok, mm, 2, 64, epochs=3, seed=4, task_sample_probs={"mir": 0.5, "captioning": 0.5}))
    assert all(torch.equal(x["input_ids"], y["input_ids"]) for x, y in zip(a, b)) and len(a) == len(b)    # seeded: reproducible


def test_task_sample_probs_are_global_not_per_rank(tmp_path):
    """the reference's repeat_shards() weights and samples the GLOBAL url list and split_by_node comes afterwards
    (m2t/data_modules.py:441-462): a rank whose r::world subset holds no shard of a task must still see that task.  Two ranks,
    shards [mir, captioning, mir, captioning]: rank 0's own subset is all-mir, rank 1's all-captioning; with weights mir 0.5 /
    captioning 0.5 both ranks read both tasks over a few epochs, the two ranks' draws are disjoint positions of ONE shared order,
    and a zero-weight task never appears on any rank."""
    names = ["mir-000", "captioning-000", "mir-001", "captioning-001"]
    for i, nm in enumerate(names):
        _make_shard(tmp_path / f"{nm}.tar", [f"{nm[0]}{i}"])
    tok = ToyTokenizer()
    tok.add_tokens(["<audio_patch>", "<audio_start>", "<audio_end>"], special_tokens=True)
    mm = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)
    pattern = ",".join(f"{tmp_path}/{nm}.tar" for nm in names)
    ids = lambda bs: {tuple(r.tolist()) for b in bs for r in b["input_ids"]}
    per_task = {}
    for task in ("mir", "captioning"):
        only = list(D.micro_batches(pattern, tok, mm, 2, 64, epochs=8, seed=2, task_sample_probs={task: 1.0, "mir" if task != "mir" else "captioning": 0.0}))
        per_task[task] = ids(only)
    common = per_task["mir"] & per_task["captioning"]                      # the "tempo ?" pair every clip carries
    mir_only, cap_only = per_task["mir"] - common, per_task["captioning"] - common
    assert mir_only and cap_only
    for rank in (0, 1):
        got = ids(D.micro_batches(pattern, tok, mm, 2, 64, rank=rank, world=2, epochs=12, seed=2, task_sample_probs={"mir": 0.5, "captioning": 0.5}))
        assert got & mir_only and got & cap_only, f"rank {rank} never saw one of the tasks"
        zero = ids(D.micro_batches(pattern, tok, mm, 2, 64, rank=rank, world=2, epochs=6, seed=2, task_sample_probs={"mir": 1.0, "captioning": 0.0}))
        assert zero and not (zero & cap_only), f"rank {rank} read a zero-weight task"
