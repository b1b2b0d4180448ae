"""CPU: pins oracle/llama_ref.py against golden vectors produced by the REAL reference wrapper
(tests/golden/llama_tiny.npz <- tests/golden/make_llama_golden.py importing
/root/reference/m2t/models/llamav2.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import llama_ref as LR

GOLD_DIR = os.path.join(os.path.dirname(__file__), "golden")


def load_gold(name):
    z = np.load(os.path.join(GOLD_DIR, name))
    s = z["spec"]
    spec = LR.LlamaSpec(hidden_size=int(s[0]), intermediate_size=int(s[1]), num_hidden_layers=int(s[2]),
                        num_attention_heads=int(s[3]), vocab_size=int(s[4]), mm_hidden_size=int(s[5]),
                        audio_start_token=int(s[6]), audio_end_token=int(s[7]), audio_patch_token=int(s[8]))
    w = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    return z, spec, w


@pytest.fixture(scope="module", params=["llama_tiny.npz", "llama_hd128.npz"])
def gold(request):
    return load_gold(request.param)


def _unused():
    z = None
    s = z["spec"]
    spec = LR.LlamaSpec(hidden_size=int(s[0]), intermediate_size=int(s[1]), num_hidden_layers=int(s[2]),
                        num_attention_heads=int(s[3]), vocab_size=int(s[4]), mm_hidden_size=int(s[5]),
                        audio_start_token=int(s[6]), audio_end_token=int(s[7]), audio_patch_token=int(s[8]))
    w = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    return z, spec, w


def test_forward_logits_and_loss(gold):
    z, spec, w = gold
    r = LR.forward(w, spec, torch.from_numpy(z["c1_ids"]), torch.from_numpy(z["c1_audio"]),
                   labels=torch.from_numpy(z["c1_labels"]))
    assert np.abs(r["logits"].numpy() - z["c1_logits"]).max() <= 2e-5
    assert abs(r["loss"].item() - float(z["c1_loss"])) <= 1e-5
    r2 = LR.forward(w, spec, torch.from_numpy(z["c1_ids"]), [torch.from_numpy(a) for a in z["c1_audio"]])
    assert np.abs(r2["logits"].numpy() - z["c2_logits"]).max() <= 2e-5


def test_text_only(gold):
    z, spec, w = gold
    r = LR.forward(w, spec, torch.from_numpy(z["c3_ids"]))
    assert np.abs(r["logits"].numpy() - z["c3_logits"]).max() <= 2e-5


def test_greedy_generate_tokens_exact(gold):
    z, spec, w = gold
    ids = LR.greedy_generate(w, spec, torch.from_numpy(z["c4_ids"]), torch.from_numpy(z["c4_audio"]), 6)
    assert np.array_equal(ids.numpy(), z["c4_generated"])
    # per-step logits through the KV cache
    out = LR.forward(w, spec, torch.from_numpy(z["c4_ids"]), torch.from_numpy(z["c4_audio"]))
    assert np.abs(out["logits"][:, -1].numpy() - z["c4_step_logits"][0]).max() <= 2e-5
    nxt = out["logits"][:, -1].argmax(-1, keepdim=True)
    out2 = LR.forward(w, spec, nxt, None, past_key_values=out["past_key_values"])
    assert np.abs(out2["logits"][:, -1].numpy() - z["c4_step_logits"][1]).max() <= 2e-5


def test_error_behaviour(gold):
    z, spec, w = gold
    aud = torch.from_numpy(z["c1_audio"])
    with pytest.raises(ValueError) as e:
        LR.forward(w, spec, torch.from_numpy(z["c5_bad_ids"]), aud)
    assert str(e.value) == str(z["c5_count_msg"])
    with pytest.raises(ValueError) as e:
        LR.forward(w, spec, torch.from_numpy(z["c5_bad2_ids"]), aud)
    assert str(e.value) == str(z["c5_follow_msg"])


def test_bf16_flow_is_close_to_fp32(gold):
    """The bf16 rounding-point flow (what the HIP path computes) stays within bf16-class distance of fp32."""
    z, spec, w = gold
    wb = {k: v.bfloat16().float() for k, v in w.items()}
    a = LR.forward(wb, spec, torch.from_numpy(z["c1_ids"]), torch.from_numpy(z["c1_audio"]))["logits"]
    b = LR.forward(wb, spec, torch.from_numpy(z["c1_ids"]), torch.from_numpy(z["c1_audio"]), act_dtype=torch.bfloat16)["logits"]
    rel = ((a - b).abs().max() / a.abs().max()).item()
    assert rel < 3e-2, rel
