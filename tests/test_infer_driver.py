"""the `.npy` representation contract and the batched / fused inference drivers.
The batched drivers must return, per example, exactly what the reference's one-example-at-a-time loop
(scripts/inference/infer_from_encodings.py:73-108 -> m2t/infer.py:infer_with_prompt) returns."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from toy_tokenizer import ToyTokenizer  # noqa: E402

PROMPT = "Describe the tempo of this clip ."
MM_CFG = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)


def _tok():
    from llark_amd.m2t.prompting import DEFAULT_CONVERSATION_HEADER
    tok = ToyTokenizer()
    for text in (DEFAULT_CONVERSATION_HEADER, "### Human: Assistant: <empty> \n " + PROMPT):
        tok.encode(text)
    return tok


def test_npy_contract_and_prompt_ids(tmp_path):
    from llark_amd.m2t import prompting as P
    from llark_amd.m2t.infer_driver import build_prompt_ids, load_encoding
    rep = np.random.default_rng(0).standard_normal((240, 4800)).astype(np.float32)
    np.save(tmp_path / "a.npy", rep)                                   # what jukebox/main.py:254 writes
    got = load_encoding(str(tmp_path / "a.npy"))
    assert got.dtype == np.float32 and got.flags["C_CONTIGUOUS"] and got.shape == (240, 4800) and np.array_equal(got, rep)
    np.save(tmp_path / "g.npy", rep.mean(0))                           # --pool-frames-per-second 0: one global frame
    assert load_encoding(str(tmp_path / "g.npy")).shape == (1, 4800)
    np.save(tmp_path / "f.npy", np.asfortranarray(rep))                # Fortran order on disk is normalised to C order
    assert load_encoding(str(tmp_path / "f.npy")).flags["C_CONTIGUOUS"]
    np.save(tmp_path / "d.npy", rep.astype(np.float64))
    with pytest.raises(ValueError, match="float32"):
        load_encoding(str(tmp_path / "d.npy"))
    np.save(tmp_path / "w.npy", rep[:, :100])
    with pytest.raises(ValueError, match="shape"):
        load_encoding(str(tmp_path / "w.npy"))
    # the prompt ids are those infer_with_prompt builds: header, human turn with <audio_start> + patches + <audio_end>
    tok = _tok()
    tok.add_tokens(["<audio_patch>", "<audio_start>", "<audio_end>"], special_tokens=True)
    end_seq = tok("\n### Assistant:").input_ids[1:]
    ids = build_prompt_ids(PROMPT, 7, tok, MM_CFG, end_seq, audio_first=True)
    enc = torch.zeros(7, 4)
    elem = {"audio_encoding": enc, "audio_encoding_shape": [7, 4], "example_id": None, "id": None,
            "conversations": [{"from": "human", "value": P.concat_audio_token_and_prompt(PROMPT, True)}, {"from": "gpt", "value": "<empty>"}]}
    elem = P.preprocess_for_lm_mappable(P.preprocess_multimodal_mappable(elem, MM_CFG), tokenizer=tok)
    assert torch.equal(ids, P.extract_prompt_tokens(elem["input_ids"], end_seq))
    patch = tok.convert_tokens_to_ids(["<audio_patch>"])[0]
    assert int((ids == patch).sum()) == 7 and ids.tolist()[-len(end_seq):] == list(end_seq)


def _tiny_model(tok, mm_hidden, max_batch):
    from llark_amd.m2t.llamav2 import WrappedLlamav2Config, WrappedLlamav2ForCausalLM
    torch.manual_seed(0)
    cfg = WrappedLlamav2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                               vocab_size=len(tok), max_position_embeddings=512, rms_norm_eps=1e-5, tie_word_embeddings=False)
    cfg.mm_hidden_size = mm_hidden
    m = WrappedLlamav2ForCausalLM(cfg).eval()
    m.get_model().initialize_adapter_modules()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_((p * 4).bfloat16().float())
    m.initialize_audio_tokenizer(mm_use_audio_start_end=True, tokenizer=tok, device="cpu")
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.bfloat16().float())
    m.cuda()
    m.configure_engine(max_batch=max_batch, max_seq=160)
    return m


@pytest.mark.gpu
def test_infer_from_encodings_batched_equals_per_example(tmp_path, monkeypatch):
    import pandas as pd
    from llark_amd.m2t import infer_driver as D
    from llark_amd.m2t.infer import infer_with_prompt
    from llark_amd.m2t.prompting import extract_response_tokens
    tok = _tok()
    m = _tiny_model(tok, 96, 3)
    end_seq = tok("\n### Assistant:").input_ids[1:]
    rng = np.random.default_rng(1)
    d = tmp_path / "reps"
    d.mkdir()
    shapes = {"c": 5, "a": 5, "e": 7, "b": 5, "d": 5}                  # 4 examples of 5 frames (batches of 3 + 1), 1 of 7
    for name, frames in shapes.items():
        np.save(d / f"{name}.npy", rng.standard_normal((frames, 96)).astype(np.float32))
    out_csv = tmp_path / "out" / "res.csv"
    recs = D.infer_from_encodings(m, tok, str(d), PROMPT, MM_CFG, end_seq, outfile=str(out_csv), batch_size=3, max_new_tokens=10)
    assert [os.path.basename(r["example_id"]) for r in recs] == ["a", "b", "c", "d", "e"]           # sorted file order
    m.configure_engine(max_batch=1, max_seq=160)
    for r in recs:
        enc = torch.from_numpy(np.load(r["example_id"] + ".npy"))
        one = infer_with_prompt(PROMPT, model=m, audio_encoding=enc, end_seq=end_seq, multimodal_cfg=MM_CFG, tokenizer=tok,
                                audio_first=True, max_new_tokens=10).cpu()
        assert r["model_completion_text"] == tok.decode(extract_response_tokens(one[0], end_seq)), r["example_id"]
        assert r["prompt_text"] == PROMPT
    df = pd.read_csv(out_csv)
    assert list(df.columns) == ["example_id", "prompt_text", "model_completion_text"] and len(df) == 5


@pytest.mark.gpu
def test_infer_from_audio_fused_equals_two_stage():
    """audio -> WrappedAudioEncoder -> generate in one process == the two-stage pipeline per clip: ``get_acts_from_audio`` (what
    jukebox/main.py writes to ``<name>.npy``: a clip shorter than the window keeps only the frames of its own audio, main.py:147)
    -> infer_with_prompt."""
    from llark_amd.jukebox import extract as E
    from llark_amd.jukebox.hparams import hparams_tiny
    from llark_amd.jukebox.synthetic import make_jukebox_weights, synthetic_clip
    from llark_amd.m2t import infer_driver as D
    from llark_amd.m2t.infer import infer_with_prompt
    from llark_amd.m2t.prompting import extract_response_tokens
    hps = hparams_tiny()
    enc = E.WrappedAudioEncoder(hps=hps, weights=make_jukebox_weights(hps, seed=0, device="cuda"), device="cuda")
    tok = _tok()
    m = _tiny_model(tok, hps.prior_width, 2)
    end_seq = tok("\n### Assistant:").input_ids[1:]
    clips = [(f"clip{i}", synthetic_clip(i, seconds=1.2 + 0.3 * i)) for i in range(3)]        # ragged lengths: padded / truncated
    recs = D.infer_from_audio(enc, m, tok, clips, PROMPT, MM_CFG, end_seq, batch_size=2, max_new_tokens=8)
    assert [r["example_id"] for r in recs] == ["clip0", "clip1", "clip2"]
    m.configure_engine(max_batch=1, max_seq=160)
    for (name, audio), r in zip(clips, recs):
        rep = torch.from_numpy(E.get_acts_from_audio(E._normalize(audio).astype(np.float32), hps, enc.vqvae, enc.top_prior, meanpool=True,
                                                     pool_frames_per_second=enc.pool_frames_per_second))
        one = infer_with_prompt(PROMPT, model=m, audio_encoding=rep, end_seq=end_seq, multimodal_cfg=MM_CFG, tokenizer=tok,
                                audio_first=True, max_new_tokens=8).cpu()
        assert r["model_completion_text"] == tok.decode(extract_response_tokens(one[0], end_seq)), name


@pytest.mark.gpu
def test_infer_from_clap_style_encodings_with_mpt(tmp_path):
    """configs[4] inference surface: (1, 512)-style CLAP embeddings (.npy, here 64-d) + the MPT backbone through the same
    batched driver; equals the per-example generate of the wrapper."""
    from llark_amd.m2t import infer_driver as D
    from llark_amd.m2t.mpt import WrappedMPTConfig, WrappedMPTForCausalLM
    from llark_amd.m2t.prompting import extract_response_tokens
    tok = _tok()
    torch.manual_seed(0)
    m = WrappedMPTForCausalLM(WrappedMPTConfig(d_model=256, n_heads=2, n_layers=2, expansion_ratio=4, max_seq_len=256, vocab_size=len(tok),
                                              mm_hidden_size=64)).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_((torch.randn_like(p) * 0.08).bfloat16().float() if p.dim() > 1 else torch.ones_like(p))
    m.initialize_audio_tokenizer(True, tok, "cpu")
    m.cuda()
    m.configure_engine(max_batch=3, max_seq=200)
    end_seq = tok("\n### Assistant:").input_ids[1:]
    rng = np.random.default_rng(2)
    d = tmp_path / "clap"
    d.mkdir()
    for name in "abcd":
        np.save(d / f"{name}.npy", rng.standard_normal((1, 64)).astype(np.float32))        # ONE frame per clip
    recs = D.infer_from_encodings(m, tok, str(d), PROMPT, MM_CFG, end_seq, batch_size=3, max_new_tokens=6)
    assert len(recs) == 4
    m.configure_engine(max_batch=1, max_seq=200)
    ids = D.build_prompt_ids(PROMPT, 1, tok, MM_CFG, end_seq)[None].cuda()
    for r in recs:
        enc = torch.from_numpy(np.load(r["example_id"] + ".npy"))[None].cuda()
        one = D.generate_batch(m, ids, enc, tok, 6)[0]
        assert r["model_completion_text"] == tok.decode(extract_response_tokens(one, end_seq)), r["example_id"]


@pytest.mark.gpu
def test_infer_cli_end_to_end(tmp_path):
    """`python -m llark_amd.m2t.infer_driver` (flags of scripts/inference/infer_from_encodings.py) on a tiny HF-format
    checkpoint + tokenizer saved locally: loads, sets up the audio tokens, generates for every .npy, writes the CSV."""
    import pandas as pd
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from llark_amd.m2t import infer_driver as D
    from llark_amd.m2t.llamav2 import WrappedLlamav2Config, WrappedLlamav2ForCausalLM
    from llark_amd.m2t.prompting import DEFAULT_CONVERSATION_HEADER
    words = sorted(set((DEFAULT_CONVERSATION_HEADER + " ### Human: Assistant: <empty> " + PROMPT).split()))
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2, "\n": 3}
    for wd in words:
        vocab.setdefault(wd, len(vocab))
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    ckpt = tmp_path / "ckpt"
    PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>").save_pretrained(str(ckpt))
    cfg = WrappedLlamav2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                               vocab_size=len(vocab), max_position_embeddings=256, rms_norm_eps=1e-5, tie_word_embeddings=False)
    cfg.mm_hidden_size = 96
    torch.manual_seed(0)
    WrappedLlamav2ForCausalLM(cfg).save_pretrained(str(ckpt))
    reps = tmp_path / "reps"
    reps.mkdir()
    rng = np.random.default_rng(3)
    for name in ("x", "y", "z"):
        np.save(reps / f"{name}.npy", rng.standard_normal((4, 96)).astype(np.float32))
    out = tmp_path / "res" / "infer.csv"
    recs = D.main(["--model_name_or_path", str(ckpt), "--audio-encodings-dir", str(reps), "--prompt", PROMPT, "--outfile", str(out),
                   "--max_new_tokens", "5", "--batch-size", "2", "--mm_hidden_size", "96", "--model_max_length", "128", "--ckpt-num", "7"])
    df = pd.read_csv(out)
    assert len(recs) == 3 and list(df.columns) == ["example_id", "prompt_text", "model_completion_text"]
    assert [os.path.basename(e) for e in df["example_id"]] == ["x", "y", "z"] and (df["prompt_text"] == PROMPT).all()
