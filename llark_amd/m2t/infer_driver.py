"""Batched inference drivers.

The reference's ``scripts/inference/infer_from_encodings.py:47-116`` walks a directory of ``<id>.npy`` Jukebox
representations (the files ``jukebox/main.py:254`` writes: float32, C-order, ``(frames, 4800)``) ONE example at a time
through ``infer_with_prompt`` and writes a CSV with the columns ``example_id, prompt_text, model_completion_text``.
Here the same contract is served in batches on the HIP engine:

* :func:`infer_from_encodings` -- same inputs / same CSV, ``batch_size`` examples per ``generate`` call;
* :func:`infer_from_audio`     -- audio clips -> ``WrappedAudioEncoder`` -> LLM in ONE process: the ``(B, frames, 4800)``
  embeddings never leave HBM (no ``.npy`` round trip).

Batched greedy decoding is token-for-token the per-example loop of the reference: every kernel on the path is
row-independent and accumulates each output element in the same k order whatever the batch size, and stopping is
tracked per sequence (``KeywordsStoppingCriteria`` semantics of ``m2t/generate.py:31-44`` applied row by row).
"""
from __future__ import annotations

import glob
import os
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .generate import KeywordsStoppingCriteria
from .prompting import (DEFAULT_CONVERSATION_HEADER, concat_audio_token_and_prompt, extract_prompt_tokens,
                        extract_response_tokens, preprocess_for_lm_mappable, preprocess_multimodal_mappable)

EMBED_DIM = 4800


def load_encoding(path: str, embed_dim: int = EMBED_DIM) -> np.ndarray:
    """One ``.npy`` representation under the contract of ``jukebox/main.py:254`` / ``m2t/gcs_utils.py:175-248``:
    float32, C-order, ``(frames, embed_dim)`` (or ``(embed_dim,)`` for the frame-rate-0 pooling -> one frame)."""
    a = np.load(path, allow_pickle=False)
    if a.dtype != np.float32:
        raise ValueError(f"{path}: expected float32 encodings, got {a.dtype}")
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2 or a.shape[1] != embed_dim:
        raise ValueError(f"{path}: expected shape (frames, {embed_dim}), got {a.shape}")
    return np.ascontiguousarray(a)


def build_prompt_ids(prompt: str, frames: int, tokenizer, multimodal_cfg: Dict[str, Any], end_seq: Sequence[int],
                     audio_first: bool = True, header: str = DEFAULT_CONVERSATION_HEADER) -> torch.Tensor:
    """Token ids of header + human turn (with the audio placeholder expanded to ``frames`` patches) up to and including
    "### Assistant:" -- exactly what ``infer_with_prompt`` (m2t/infer.py:99-145) feeds to ``generate``."""
    text = concat_audio_token_and_prompt(prompt, audio_first)
    elem = {"audio_encoding": np.zeros((frames, 1), dtype=np.float32), "audio_encoding_shape": [frames, 1], "id": None,
            "conversations": [{"from": "human", "value": text}, {"from": "gpt", "value": "<empty>"}]}
    elem = preprocess_for_lm_mappable(preprocess_multimodal_mappable(elem, multimodal_cfg), tokenizer=tokenizer, header=header)
    return extract_prompt_tokens(elem["input_ids"], end_seq)


@torch.no_grad()
def generate_batch(model, input_ids: torch.Tensor, encodings: torch.Tensor, tokenizer, max_new_tokens: int = 512,
                   keywords: Sequence[str] = ("###",)) -> List[torch.Tensor]:
    """Greedy generation for B examples sharing one prompt length.  Returns, per example, prompt + continuation ids cut
    where the reference's loop would have stopped for that example (keyword hit, EOS, or ``max_new_tokens``)."""
    B = input_ids.shape[0]
    dev = model.engine.device if getattr(model, "_engine", None) is not None else torch.device("cuda")
    ids = input_ids.to(dev)
    stops = [KeywordsStoppingCriteria(keywords=list(keywords), tokenizer=tokenizer, input_ids=input_ids[b: b + 1]) for b in range(B)]
    eos = getattr(getattr(model, "generation_config", None), "eos_token_id", None)
    done_at: List[Optional[int]] = [None] * B

    class _AllDone:                                        # plugs the per-row bookkeeping into model.generate's loop
        def __call__(self, out_ids, scores=None, **kw):
            n = out_ids.shape[1]
            rows = out_ids.cpu()
            for b in range(B):
                if done_at[b] is None and (stops[b](rows[b: b + 1], None) or (eos is not None and int(rows[b, -1]) == eos)):
                    done_at[b] = n
            return all(d is not None for d in done_at)

    out = model.generate(input_ids=ids, audio_encodings=encodings, max_new_tokens=max_new_tokens, stopping_criteria=[_AllDone()],
                         eos_token_id=-1)                  # EOS is handled per row above
    out = out.cpu()
    return [out[b, : (done_at[b] if done_at[b] is not None else out.shape[1])] for b in range(B)]


def _rows_to_records(example_ids, prompt, outs, end_seq, tokenizer) -> List[Dict[str, str]]:
    recs = []
    for ex, ids in zip(example_ids, outs):
        completion = tokenizer.decode(extract_response_tokens(ids, end_seq))
        recs.append({"example_id": ex, "prompt_text": prompt, "model_completion_text": completion})
    return recs


def _write_csv(records: List[Dict[str, str]], outfile: Optional[str]) -> None:
    if not outfile:
        return
    import pandas as pd

    d = os.path.dirname(outfile)
    if d and not os.path.exists(d):
        os.makedirs(d)
    pd.DataFrame(records, columns=["example_id", "prompt_text", "model_completion_text"]).to_csv(outfile, index=False)


def infer_from_encodings(model, tokenizer, audio_encodings_dir: str, prompt: str, multimodal_cfg: Dict[str, Any],
                         end_seq: Sequence[int], outfile: Optional[str] = None, batch_size: int = 8,
                         max_samples: Optional[int] = None, max_new_tokens: int = 512, audio_first: bool = True,
                         embed_dim: Optional[int] = None) -> List[Dict[str, str]]:
    """Directory of ``*.npy`` -> CSV, like ``scripts/inference/infer_from_encodings.py:main`` (same record fields; files
    in sorted order; ``example_id`` = path without ``.npy``).  Examples are grouped by frame count so that every batch
    shares one prompt length."""
    if embed_dim is None:                                  # 4800 for Jukebox, 512 for CLAP (ModelArguments.mm_hidden_size)
        embed_dim = int(getattr(getattr(model, "config", None), "mm_hidden_size", EMBED_DIM))
    paths = sorted(glob.glob(os.path.join(audio_encodings_dir, "*.npy")))
    if max_samples:
        paths = paths[:max_samples]
    by_frames: Dict[int, List[Tuple[str, np.ndarray]]] = {}
    for p in paths:
        a = load_encoding(p, embed_dim)
        by_frames.setdefault(a.shape[0], []).append((p, a))
    records: Dict[str, Dict[str, str]] = {}
    for frames, items in by_frames.items():
        prompt_ids = build_prompt_ids(prompt, frames, tokenizer, multimodal_cfg, end_seq, audio_first)
        for i in range(0, len(items), batch_size):
            chunk = items[i: i + batch_size]
            enc = torch.from_numpy(np.stack([a for _, a in chunk])).cuda()
            ids = prompt_ids.unsqueeze(0).repeat(len(chunk), 1)
            outs = generate_batch(model, ids, enc, tokenizer, max_new_tokens)
            for rec in _rows_to_records([p[: -len(".npy")] for p, _ in chunk], prompt, outs, end_seq, tokenizer):
                records[rec["example_id"]] = rec
    ordered = [records[p[: -len(".npy")]] for p in paths]
    _write_csv(ordered, outfile)
    return ordered


def infer_from_audio(encoder, model, tokenizer, clips: Iterable[Tuple[str, np.ndarray]], prompt: str,
                     multimodal_cfg: Dict[str, Any], end_seq: Sequence[int], outfile: Optional[str] = None, batch_size: int = 8,
                     max_new_tokens: int = 512, audio_first: bool = True) -> List[Dict[str, str]]:
    """Fused driver: ``clips`` yields ``(example_id, mono float32 audio @44.1 kHz)``; ``encoder`` is a
    ``llark_amd.jukebox.extract.WrappedAudioEncoder``.  Audio is normalised / padded / truncated exactly as
    ``jukebox/main.py:29-59`` does, encoded in batches on the GPU, short clips keep only the frames of their own audio
    (``jukebox/main.py:147``), and the ``(B, frames, 4800)`` tensor is handed to ``generate`` without leaving device memory."""
    from ..jukebox import extract as E

    records: List[Dict[str, str]] = []
    buf: List[Tuple[str, np.ndarray]] = []
    prompt_cache: Dict[int, torch.Tensor] = {}

    def flush():
        if not buf:
            return
        from math import floor

        hps = encoder.hps
        n = hps.sample_length
        norm = [E._normalize(a) for _, a in buf]
        audio = np.stack([E.maybe_pad_audio_to_max_len(a, n)[:n].astype(np.float32) for a in norm])
        emb = encoder(torch.from_numpy(audio).to(encoder.vqvae.device))                  # (B, frames, 4800) fp32, on device
        # jukebox/main.py:147: activations are truncated to latent_audio_len = floor(T * len / expected) BEFORE pooling, so a clip
        # shorter than 23.8 s yields fewer frames (no embeddings of the zero padding).  Group the batch by frame count so that
        # every generate call shares one prompt length, exactly as the .npy pipeline (infer_from_encodings) does.
        fl = encoder.frame_len
        frames = [min(floor(hps.n_ctx * len(a) / n), hps.n_ctx) // fl for a in norm]
        recs: Dict[int, Dict[str, str]] = {}
        for f in sorted(set(frames)):
            if f == 0:
                raise ValueError("infer_from_audio: a clip is shorter than one pooled frame (%d samples)" % (fl * hps.raw_to_tokens))
            rows = [i for i, fi in enumerate(frames) if fi == f]
            if f not in prompt_cache:
                prompt_cache[f] = build_prompt_ids(prompt, f, tokenizer, multimodal_cfg, end_seq, audio_first)
            ids = prompt_cache[f].unsqueeze(0).repeat(len(rows), 1)
            sub = emb[torch.tensor(rows, device=emb.device), :f].contiguous()
            outs = generate_batch(model, ids, sub, tokenizer, max_new_tokens)
            for i, rec in zip(rows, _rows_to_records([buf[i][0] for i in rows], prompt, outs, end_seq, tokenizer)):
                recs[i] = rec
        records.extend(recs[i] for i in range(len(buf)))
        buf.clear()

    for item in clips:
        buf.append(item)
        if len(buf) == batch_size:
            flush()
    flush()
    _write_csv(records, outfile)
    return records


def get_prompt_end_token_sequence(tokenizer, model_name: str, prompt_end_string: str = "\n### Assistant:") -> List[int]:
    """m2t/tokenizer.py:33-52: the token ids that mark the end of the prompt; the Llama-2 SentencePiece tokenizer prepends a
    piece to a string that starts with a newline, which is dropped."""
    end_seq = tokenizer([prompt_end_string], add_special_tokens=False).input_ids[0]
    if "meta-llama/Llama-2" in model_name:
        end_seq = end_seq[1:]
    return list(end_seq)


def main(argv=None):
    """CLI with the flags of the reference's ``scripts/inference/infer_from_encodings.py:120-156`` (+ ``--batch-size``):
    python -m llark_amd.m2t.infer_driver --model_name_or_path <dir> --audio-encodings-dir reps/ --prompt "What genre is this song?" \
        --outfile results/infer.csv [--max-samples N] [--max_new_tokens 512] [--batch-size 8] [--mm_hidden_size 4800]"""
    import argparse

    from transformers import AutoTokenizer

    from .llamav2 import WrappedLlamav2ForCausalLM
    from .special_tokens import DEFAULT_AUDIO_END_TOKEN, DEFAULT_AUDIO_PATCH_TOKEN, DEFAULT_AUDIO_START_TOKEN

    ap = argparse.ArgumentParser(description="LLark inference over a directory of audio encodings on the HIP engine")
    ap.add_argument("--model_name_or_path", required=True)
    ap.add_argument("--audio-encodings-dir", required=True)
    ap.add_argument("--prompt", required=True)
    ap.add_argument("--outfile", default="infer_results.csv")
    ap.add_argument("--max-samples", type=int, default=None)
    ap.add_argument("--max_new_tokens", type=int, default=512)
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--mm_hidden_size", type=int, default=None)
    ap.add_argument("--model_max_length", type=int, default=2048)
    ap.add_argument("--llm-precision", default="split", choices=["split", "bf16"])
    for ignored in ("--ckpt-num", "--report_to", "--bf16", "--tf32", "--output_dir"):
        ap.add_argument(ignored, default=None, help="accepted for script compatibility")
    args = ap.parse_args(argv)

    tok = AutoTokenizer.from_pretrained(args.model_name_or_path, model_max_length=args.model_max_length, padding_side="right", use_fast=False)
    model = WrappedLlamav2ForCausalLM.from_pretrained(args.model_name_or_path, torch_dtype=torch.bfloat16)
    if args.mm_hidden_size:
        model.config.mm_hidden_size = args.mm_hidden_size
    if tok.pad_token is None:                                           # m2t/train.py:110-124 adds it at training time
        tok.add_special_tokens(dict(pad_token="[PAD]"))
        model.resize_token_embeddings(len(tok))
    if not hasattr(model.get_model(), "mm_projector"):
        model.get_model().initialize_adapter_modules()
    known = set(tok.get_vocab())
    if {DEFAULT_AUDIO_PATCH_TOKEN, DEFAULT_AUDIO_START_TOKEN, DEFAULT_AUDIO_END_TOKEN} <= known and model.get_input_embeddings().weight.shape[0] >= len(tok):
        ac = model.get_model().audio_encoder_config                      # a trained checkpoint: the tokens are already in place
        ac.use_audio_start_end = True
        ac.audio_patch_token, ac.audio_start_token, ac.audio_end_token = tok.convert_tokens_to_ids(
            [DEFAULT_AUDIO_PATCH_TOKEN, DEFAULT_AUDIO_START_TOKEN, DEFAULT_AUDIO_END_TOKEN])
    else:
        model.initialize_audio_tokenizer(mm_use_audio_start_end=True, tokenizer=tok, device="cpu")
    model.cuda().eval()
    model.configure_engine(max_batch=args.batch_size, max_seq=args.model_max_length, precision=args.llm_precision)
    end_seq = get_prompt_end_token_sequence(tok, args.model_name_or_path)
    mm_cfg = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)
    recs = infer_from_encodings(model, tok, args.audio_encodings_dir, args.prompt, mm_cfg, end_seq, outfile=args.outfile,
                                batch_size=args.batch_size, max_samples=args.max_samples, max_new_tokens=args.max_new_tokens)
    print(f"writing {len(recs)} results to {args.outfile}")
    return recs


if __name__ == "__main__":
    main()
