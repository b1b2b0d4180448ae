"""Host logic of llark_amd.m2t.checkpoint that needs no GPU: folder discovery / ordering / pruning rules and the
adapter side-file key filter of m2t/models/trainer.py:44-48."""
import os

import torch

from llark_amd.m2t import checkpoint as CK


def test_checkpoint_discovery_and_adapter_filter(tmp_path):
    out = tmp_path / "run"
    assert CK.latest_checkpoint(str(out)) is None
    for n in (5000, 10000, 900):
        (out / f"checkpoint-{n}").mkdir(parents=True)
    (out / "checkpoint-abc").mkdir()                     # not a step folder
    (out / "checkpoint-7.bin").write_bytes(b"")          # a file, not a folder
    (out / "mm_projector").mkdir()
    got = [os.path.basename(p) for p in CK.list_checkpoints(str(out))]
    assert got == ["checkpoint-900", "checkpoint-5000", "checkpoint-10000"]           # numeric, not lexical, order
    assert os.path.basename(CK.latest_checkpoint(str(out))) == "checkpoint-10000"
    sd = {"model.layers.0.self_attn.q_proj.weight": torch.zeros(1), "model.mm_projector.weight": torch.zeros(1),
          "model.mm_projector.bias": torch.zeros(1), "model.embed_tokens.weight": torch.zeros(1), "lm_head.weight": torch.zeros(1),
          "transformer.embed_in.weight": torch.zeros(1)}
    assert sorted(CK.adapter_state(sd)) == ["model.embed_tokens.weight", "model.mm_projector.bias", "model.mm_projector.weight",
                                            "transformer.embed_in.weight"]
