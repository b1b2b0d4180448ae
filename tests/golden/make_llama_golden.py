"""Generates tests/golden/llama_*.npz by running the REAL reference wrapper
(/root/reference/m2t/models/llamav2.py over the installed transformers, fp32, eager attention, CPU).

Run in the build container only (needs /root/reference):  python tests/golden/make_llama_golden.py
The fixtures travel with the repo; nothing on the GPU box reads /root/reference.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from m2t.models.llamav2 import WrappedLlamav2Config, WrappedLlamav2ForCausalLM  # noqa: E402  (the reference)

from oracle import llama_ref as LR  # noqa: E402


def build(spec: LR.LlamaSpec, w):
    cfg = WrappedLlamav2Config(hidden_size=spec.hidden_size, intermediate_size=spec.intermediate_size,
                               num_hidden_layers=spec.num_hidden_layers, num_attention_heads=spec.num_attention_heads,
                               num_key_value_heads=spec.num_attention_heads, vocab_size=spec.vocab_size,
                               max_position_embeddings=512, rms_norm_eps=spec.rms_norm_eps, rope_theta=spec.rope_theta,
                               tie_word_embeddings=False)
    cfg.mm_hidden_size = spec.mm_hidden_size
    cfg._attn_implementation = "eager"
    m = WrappedLlamav2ForCausalLM(cfg).eval()
    m.get_model().initialize_adapter_modules()
    missing, unexpected = m.load_state_dict({k: v.float() for k, v in w.items()}, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    ac = m.get_model().audio_encoder_config
    ac.audio_start_token, ac.audio_end_token, ac.audio_patch_token = (spec.audio_start_token, spec.audio_end_token,
                                                                      spec.audio_patch_token)
    return m


def main():
    # fixture 1: tiny heads (head_dim 16) -- pins the oracle's generic arithmetic
    make(LR.LlamaSpec(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, vocab_size=100,
                      mm_hidden_size=48, audio_start_token=98, audio_end_token=99, audio_patch_token=97),
         "llama_tiny.npz", std=0.2, mm=48)
    # fixture 2: Llama-2's head_dim 128 (what the HIP attention kernels are built for)
    make(LR.LlamaSpec(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, vocab_size=100,
                      mm_hidden_size=96, audio_start_token=98, audio_end_token=99, audio_patch_token=97),
         "llama_hd128.npz", std=0.08, mm=96, bf16_valued=True)


def make(spec, fname, std, mm, bf16_valued=False):
    w = LR.make_weights(spec, seed=0, std=std)
    if bf16_valued:      # what the reference's `model.to(bf16)` holds: every parameter exactly representable in bf16
        w = {k: v.bfloat16().float() for k, v in w.items()}
    m = build(spec, w)
    g = torch.Generator().manual_seed(1)
    F_ = 5

    def ids_with_audio(n_text_before, n_text_after):
        return ([1] + torch.randint(3, 90, (n_text_before,), generator=g).tolist() + [98] + [97] * F_ + [99]
                + torch.randint(3, 90, (n_text_after,), generator=g).tolist())

    out = {f"w::{k}": v.numpy() for k, v in w.items()}
    out["spec"] = np.array([spec.hidden_size, spec.intermediate_size, spec.num_hidden_layers, spec.num_attention_heads,
                            spec.vocab_size, spec.mm_hidden_size, 98, 99, 97])
    # case 1: batch of 2, tensor audio, labels
    ids = torch.tensor([ids_with_audio(3, 8), ids_with_audio(5, 6)])
    aud = torch.randn(2, F_, mm, generator=g)
    labels = ids.clone()
    labels[:, :12] = -100
    with torch.no_grad():
        r = m(input_ids=ids, audio_encodings=aud, labels=labels)
    out.update(c1_ids=ids.numpy(), c1_audio=aud.numpy(), c1_labels=labels.numpy(), c1_logits=r.logits.numpy(),
               c1_loss=np.array(r.loss.item()))
    # case 2: list-of-tensors audio encodings
    with torch.no_grad():
        r2 = m(input_ids=ids, audio_encodings=[aud[0], aud[1]])
    out.update(c2_logits=r2.logits.numpy())
    # case 3: text only (no audio tokens), audio_encodings=None
    ids3 = torch.randint(3, 90, (2, 10), generator=g)
    with torch.no_grad():
        r3 = m(input_ids=ids3)
    out.update(c3_ids=ids3.numpy(), c3_logits=r3.logits.numpy())
    # case 4: greedy decoding.  NOT through m.generate(): under the installed transformers 5.15 the
    # reference's prepare_inputs_for_generation (m2t/models/llamav2.py:339-365, written for 4.29.2)
    # receives an empty-but-truthy DynamicCache on the first step and drops the prompt, so
    # m.generate() does not reproduce the pinned-version behaviour.  The loop below is what 4.29.2's
    # greedy_search does with that hook: full prompt (+audio) once, then the last token with the
    # cache, audio_encodings forwarded every step -- all through the reference's own forward().
    ids4 = torch.tensor([ids_with_audio(2, 4)])
    aud4 = torch.randn(1, F_, mm, generator=g)
    gen = ids4.clone()
    with torch.no_grad():
        r4 = m(input_ids=ids4, audio_encodings=aud4, use_cache=True)
        step_logits = [r4.logits[:, -1].numpy()]
        for _ in range(6):
            nxt = r4.logits[:, -1].argmax(-1, keepdim=True)
            gen = torch.cat((gen, nxt), dim=1)
            r4 = m(input_ids=nxt, past_key_values=r4.past_key_values, use_cache=True, audio_encodings=aud4)
            step_logits.append(r4.logits[:, -1].numpy())
    out.update(c4_ids=ids4.numpy(), c4_audio=aud4.numpy(), c4_generated=gen.numpy(), c4_step_logits=np.stack(step_logits))
    # case 5: error behaviour
    bad = ids.clone()
    bad[0, (bad[0] == 99).nonzero()[0, 0]] = 5          # end token missing -> count mismatch
    try:
        m(input_ids=bad, audio_encodings=aud)
        msg = ""
    except ValueError as e:
        msg = str(e)
    out["c5_count_msg"] = np.array(msg)
    bad2 = ids.clone()
    pos = (bad2[0] == 99).nonzero()[0, 0]
    bad2[0, pos], bad2[0, pos + 1] = bad2[0, pos + 1].item(), 99   # end token one position late
    try:
        m(input_ids=bad2, audio_encodings=aud)
        msg2 = ""
    except ValueError as e:
        msg2 = str(e)
    out["c5_follow_msg"] = np.array(msg2)
    out["c5_bad_ids"] = bad.numpy()
    out["c5_bad2_ids"] = bad2.numpy()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print("wrote", fname, {k: getattr(v, "shape", None) for k, v in out.items() if not k.startswith("w::")})


if __name__ == "__main__":
    main()
