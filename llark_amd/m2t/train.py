"""Instruction-tuning loop with the hyper-parameters of the reference's ``m2t/train.py`` +
``scripts/training/train_llark.sh:20-49`` on the native HIP training step (``HipLlamaTrainer``): per-device
micro-batches, gradient accumulation, AdamW (lr 5e-5, weight decay 0), cosine schedule with 3 % linear warm-up,
data-parallel gradient all-reduce over RCCL, bf16.

Not reproduced (I/O + HF-Trainer glue out of scope): webdataset / GCS readers, wandb logging,
HF ``TrainingArguments`` parsing, LoRA / bitsandbytes / FSDP options.  Batches are the dictionaries produced by
``llark_amd.m2t.prompting.DataCollatorForSupervisedDataset`` (same keys as the reference's collator).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Iterable, Optional

import torch

from .. import dist as D
from .llamav2 import plan_audio_splice
from .train_engine import HipLlamaTrainer


@dataclass
class TrainConfig:
    learning_rate: float = 5e-5                 # train_llark.sh:26
    weight_decay: float = 0.0                   # :37
    warmup_ratio: float = 0.03                  # :36
    max_steps: int = 100000                     # :38
    gradient_accumulation_steps: int = 4        # :27
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    lr_scheduler_type: str = "cosine"           # :35
    gradient_checkpointing: bool = False        # train_llark.sh:25 (per-layer recompute in the backward; same gradients, less HBM)
    max_grad_norm: float = 1.0                  # transformers TrainingArguments default (clip_grad_norm_ in Trainer's step); not set by train_llark.sh
    fuse_accumulation: bool = False             # run the accumulation micro-batches of a step as ONE pass when they share a shape and the engine
                                                # holds them (loss normalised per micro-batch: the same gradient; 8 x 2048: 714 -> 685 ms, +60 GB)
    grad_comm: str = "bf16"                     # the reference's DDP buckets are bf16 (m2t/train.py:94-103 casts the model): 13.5 GB/step; "fp32" = 27 GB


def lr_at(step: int, cfg: TrainConfig) -> float:
    """HF ``get_cosine_schedule_with_warmup`` (step counted from 0)."""
    warm = math.ceil(cfg.max_steps * cfg.warmup_ratio)
    if step < warm:
        return cfg.learning_rate * step / max(1, warm)
    prog = (step - warm) / max(1, cfg.max_steps - warm)
    return cfg.learning_rate * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def train(engine, batches: Iterable[Dict], audio_cfg, cfg: TrainConfig = TrainConfig(), world: int = 1,
          max_optimizer_steps: Optional[int] = None, log=None, output_dir: Optional[str] = None, save_steps: int = 5000,
          save_total_limit: Optional[int] = 1, rank: int = 0, hf_config=None, tokenizer=None):
    """engine: a bf16 ``HipLlamaEngine`` holding the weights; batches: collated micro-batches of THIS rank
    (``input_ids``, ``labels``, ``attention_mask``, ``audio_encodings``).  Returns the list of logged losses.
    With ``output_dir``: resumes from the newest ``checkpoint-*`` there (m2t/train.py:257-260), saves every
    ``save_steps`` optimizer steps (train_llark.sh:31,41-42) and once more at the end; ``hf_config`` / ``tokenizer`` are
    written into every checkpoint folder so that ``from_pretrained`` / ``load_pretrained_model`` can open it.
    ``batches`` may be a callable ``skip_micro_batches -> iterable``: it is then called with the number of micro-batches
    the resumed run has already consumed (``step * gradient_accumulation_steps``) so the data stream continues."""
    toks = [t for t in (audio_cfg.audio_start_token, audio_cfg.audio_end_token) if isinstance(t, int)]
    tr = HipLlamaTrainer(engine, lr=cfg.learning_rate, betas=(cfg.adam_beta1, cfg.adam_beta2), eps=cfg.adam_epsilon,
                         weight_decay=cfg.weight_decay, embed_grad_tokens=toks,
                         grad_comm=torch.bfloat16 if cfg.grad_comm == "bf16" else torch.float32,
                         gradient_checkpointing=cfg.gradient_checkpointing)
    from . import checkpoint as CK

    if output_dir:
        CK.maybe_resume(tr, output_dir)
    if callable(batches):
        batches = batches(tr.step_count * cfg.gradient_accumulation_steps)
    if max_optimizer_steps and tr.step_count >= max_optimizer_steps:
        return []
    losses, acc, micro = [], 0.0, 0
    accum = cfg.gradient_accumulation_steps
    pending = []                                            # fuse_accumulation: the step's micro-batches until the last one arrives

    def run_micro(ids, segs, labels, last):
        loss = tr.forward_backward(ids, segs, labels, 1.0 / accum, overlap_allreduce_world=world if last else 1,
                                   last_micro_batch=last and cfg.max_grad_norm > 0)
        return float(loss.item()) / accum

    for batch in batches:
        ids = batch["input_ids"].to(engine.device)
        feats = batch.get("audio_encodings")
        if feats is not None:
            feats = [f.to(engine.device, torch.float32) for f in feats] if isinstance(feats, (list, tuple)) \
                else feats.to(engine.device, torch.float32)
        segs = plan_audio_splice(ids, feats, audio_cfg, False)
        labels = batch["labels"].to(engine.device)
        last = (micro + 1) % accum == 0                     # DDP no_sync boundary: exchange only on the last micro-step
        if cfg.fuse_accumulation and accum > 1:
            pending.append((ids, segs, labels))
            if last:
                B = ids.shape[0]
                if len({tuple(p[0].shape) for p in pending}) == 1 and B * accum <= engine.max_batch:
                    all_segs = [(b + k * B, st, fr) for k, p in enumerate(pending) for (b, st, fr) in p[1]]
                    loss = tr.forward_backward(torch.cat([p[0] for p in pending]), all_segs, torch.cat([p[2] for p in pending]), 1.0 / accum,
                                               overlap_allreduce_world=world, last_micro_batch=cfg.max_grad_norm > 0, loss_groups=accum)
                    acc = float(loss.item())
                else:                                       # ragged step (the collator pads each micro-batch to its own longest sequence)
                    for k, p in enumerate(pending):
                        acc += run_micro(*p, k == accum - 1)
                pending = []
        else:
            acc += run_micro(ids, segs, labels, last)
        micro += 1
        if micro % accum == 0:
            tr.allreduce_grads(world)                       # the one exchange step of the path
            tr.lr = lr_at(tr.step_count, cfg)
            tr.step(world, max_grad_norm=cfg.max_grad_norm)
            mean_loss = D.max_over_ranks(acc, 1)            # local value; rank 0 logs
            losses.append(mean_loss)
            if log:
                log(dict(step=tr.step_count, loss=mean_loss, lr=tr.lr))
            acc = 0.0
            if output_dir and tr.step_count % save_steps == 0:
                CK.save_checkpoint(tr, output_dir, save_total_limit, rank, hf_config, tokenizer)
            if max_optimizer_steps and tr.step_count >= max_optimizer_steps:
                break
    if output_dir:
        CK.save_checkpoint(tr, output_dir, save_total_limit, rank, hf_config, tokenizer)
    return losses


def check_supported_recipe(args) -> None:
    """Flags whose reference meaning the native loop does NOT implement must not be swallowed: a run that silently
    trains something else than asked is worse than no run.
      --freeze_backbone True  : m2t/train.py:150-151 ``model.requires_grad_(False)`` (adapter pre-training) -- the HIP
                                trainer always updates the backbone;
      --tune_mm_mlp_adapter False: the reference then leaves EVERY embedding row and lm_head trainable
                                (llamav2.py:395-419); the HIP trainer trains the two audio-token rows and freezes lm_head;
      --lr_scheduler_type     : only HF's cosine-with-warmup is implemented (train_llark.sh:35);
      --bf16 False            : the HIP training step is the bf16 flow."""
    if getattr(args, "freeze_backbone", False):
        raise NotImplementedError("--freeze_backbone True (adapter-only pre-training) is not implemented by the HIP trainer; "
                                  "it would fine-tune all backbone weights")
    if not getattr(args, "tune_mm_mlp_adapter", True):
        raise NotImplementedError("--tune_mm_mlp_adapter False (all embedding rows + lm_head trainable) is not implemented: the HIP "
                                  "trainer updates the audio-token embedding rows only and keeps lm_head frozen")
    if getattr(args, "lr_scheduler_type", "cosine") != "cosine":
        raise NotImplementedError(f"--lr_scheduler_type {args.lr_scheduler_type!r}: only 'cosine' (with warm-up) is implemented")
    if not getattr(args, "bf16", True):
        raise NotImplementedError("--bf16 False: the HIP training step computes in the bf16 flow of the reference recipe")


def main(argv=None):
    """CLI with the flag names of ``m2t/train.py`` / ``scripts/training/train_llark.sh`` that the native loop implements:
    python -m llark_amd.m2t.train --model_name_or_path <hf dir> --train_data_path 'shards-{000..127}.tar' --output_dir out \
        --mm_hidden_size 4800 --mm_use_audio_start_end True --per_device_train_batch_size 2 --gradient_accumulation_steps 4 \
        --learning_rate 5e-5 --warmup_ratio 0.03 --max_steps 100000 --model_max_length 2048 --save_steps 5000 --save_total_limit 1
    (one process per GPU under ``python -m torch.distributed.run``)."""
    import argparse

    import torch as _torch
    from transformers import AutoTokenizer

    from .. import dist as D
    from .data import micro_batches
    from .engine import HipLlamaEngine, LlamaDims
    from .llamav2 import WrappedLlamav2ForCausalLM

    boolean = lambda v: str(v).lower() in ("1", "true", "yes")
    ap = argparse.ArgumentParser(description="LLark instruction tuning on the HIP training step")
    ap.add_argument("--model_name_or_path", required=True)
    ap.add_argument("--train_data_path", required=True)
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--mm_hidden_size", type=int, default=4800)
    ap.add_argument("--mm_use_audio_start_end", type=boolean, default=True)
    ap.add_argument("--tune_mm_mlp_adapter", type=boolean, default=True)
    ap.add_argument("--per_device_train_batch_size", type=int, default=2)
    ap.add_argument("--gradient_accumulation_steps", type=int, default=4)
    ap.add_argument("--learning_rate", type=float, default=5e-5)
    ap.add_argument("--weight_decay", type=float, default=0.0)
    ap.add_argument("--warmup_ratio", type=float, default=0.03)
    ap.add_argument("--max_steps", type=int, default=100000)
    ap.add_argument("--model_max_length", type=int, default=2048)
    ap.add_argument("--save_steps", type=int, default=5000)
    ap.add_argument("--save_total_limit", type=int, default=1)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--grad_comm", default="bf16", choices=["fp32", "bf16"], help="transport dtype of the gradient all-reduce (reference: bf16)")
    ap.add_argument("--allow_pickle", type=boolean, default=False, help=".pyd shard members are pickles: enable only for trusted data")
    ap.add_argument("--fuse_accumulation", type=boolean, default=False,
                    help="not a reference flag: run the accumulation micro-batches of a step as one pass when they share a shape (same gradient; more HBM)")
    ap.add_argument("--gradient_checkpointing", type=boolean, default=False, help="train_llark.sh:25: recompute each decoder layer in the backward")
    ap.add_argument("--apply_task_sample_probs", type=boolean, default=False, help="m2t/arguments.py:68: weight shards by the task in their name")
    ap.add_argument("--task_sample_probs", default=None, help='JSON dict task -> probability (default: m2t/arguments.py:61-67)')
    ap.add_argument("--freeze_backbone", type=boolean, default=False)
    ap.add_argument("--lr_scheduler_type", default="cosine")
    ap.add_argument("--bf16", type=boolean, default=True)
    ap.add_argument("--max_grad_norm", type=float, default=1.0, help="HF TrainingArguments.max_grad_norm (gradient clipping; 0 = off)")
    for ignored in ("--tf32", "--report_to", "--logging_steps", "--evaluation_strategy", "--save_strategy",
                    "--ddp_find_unused_parameters", "--dataloader_num_workers", "--num_train_epochs"):
        ap.add_argument(ignored, default=None, help="accepted for script compatibility (no effect on the arithmetic of the native loop)")
    args = ap.parse_args(argv)

    check_supported_recipe(args)
    rank, world, local = D.env_rank_world()
    _torch.cuda.set_device(local)
    dev = _torch.device("cuda", local)
    if world > 1:
        D.init(backend="nccl", device=dev)
    tok = AutoTokenizer.from_pretrained(args.model_name_or_path, model_max_length=args.model_max_length, padding_side="right", use_fast=False)
    if tok.pad_token is None:
        tok.add_special_tokens(dict(pad_token="[PAD]"))                    # m2t/train.py:110-124
    model = WrappedLlamav2ForCausalLM.from_pretrained(args.model_name_or_path, torch_dtype=_torch.bfloat16)
    model.config.mm_hidden_size = args.mm_hidden_size
    model.get_model().initialize_adapter_modules(tune_mm_mlp_adapter=args.tune_mm_mlp_adapter)
    model.resize_token_embeddings(len(tok))
    model.initialize_audio_tokenizer(args.mm_use_audio_start_end, tok, device=dev, tune_mm_mlp_adapter=args.tune_mm_mlp_adapter)
    c = model.config
    dims = LlamaDims(hidden_size=c.hidden_size, intermediate_size=c.intermediate_size, num_hidden_layers=c.num_hidden_layers,
                     num_attention_heads=c.num_attention_heads, vocab_size=len(tok), rms_norm_eps=c.rms_norm_eps,
                     rope_theta=getattr(c, "rope_theta", 10000.0), mm_hidden_size=args.mm_hidden_size)
    eng = HipLlamaEngine(dims, dev, max_batch=args.per_device_train_batch_size * (args.gradient_accumulation_steps if args.fuse_accumulation else 1),
                         max_seq=args.model_max_length, precision="bf16", frag_weights=False)
    eng.load_state_dict(model.state_dict())
    audio_cfg = model.get_model().audio_encoder_config
    hf_config = model.config                                              # written into every checkpoint-N (with the tokenizer)
    hf_config.mm_hidden_size = args.mm_hidden_size
    hf_config.tune_mm_mlp_adapter = args.tune_mm_mlp_adapter
    hf_config.mm_use_audio_start_end = args.mm_use_audio_start_end
    hf_config.vocab_size = len(tok)
    del model
    mm_cfg = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=args.mm_use_audio_start_end)
    cfg = TrainConfig(learning_rate=args.learning_rate, weight_decay=args.weight_decay, warmup_ratio=args.warmup_ratio,
                      max_steps=args.max_steps, gradient_accumulation_steps=args.gradient_accumulation_steps, grad_comm=args.grad_comm,
                      gradient_checkpointing=args.gradient_checkpointing, max_grad_norm=args.max_grad_norm,
                      fuse_accumulation=args.fuse_accumulation)
    task_probs = None
    if args.apply_task_sample_probs:
        import json as _json

        from .data import DEFAULT_TASK_SAMPLE_PROBS

        task_probs = _json.loads(args.task_sample_probs) if args.task_sample_probs else dict(DEFAULT_TASK_SAMPLE_PROBS)

    def batches(skip_micro_batches: int):
        return micro_batches(args.train_data_path, tok, mm_cfg, args.per_device_train_batch_size, args.model_max_length, rank, world,
                             seed=args.seed, allow_pickle=args.allow_pickle, skip_micro_batches=skip_micro_batches,
                             task_sample_probs=task_probs)

    log = (lambda rec: print(rec, flush=True)) if rank == 0 else None
    train(eng, batches, audio_cfg, cfg, world=world, max_optimizer_steps=args.max_steps, log=log, output_dir=args.output_dir,
          save_steps=args.save_steps, save_total_limit=args.save_total_limit, rank=rank, hf_config=hf_config, tokenizer=tok)
    D.shutdown(world)


if __name__ == "__main__":
    main()
