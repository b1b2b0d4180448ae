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


def lr_at(step: int, cfg: TrainConfig) -> float:
    """HF ``get_cosine_schedule_with_warmup`` (step counted from 0)."""
    warm = math.ceil(cfg.max_steps * cfg.warmup_ratio)
    if step < warm:
        return cfg.learning_rate * step / max(1, warm)
    prog = (step - warm) / max(1, cfg.max_steps - warm)
    return cfg.learning_rate * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def train(engine, batches: Iterable[Dict], audio_cfg, cfg: TrainConfig = TrainConfig(), world: int = 1,
          max_optimizer_steps: Optional[int] = None, log=None, output_dir: Optional[str] = None, save_steps: int = 5000,
          save_total_limit: Optional[int] = 1, rank: int = 0):
    """engine: a bf16 ``HipLlamaEngine`` holding the weights; batches: collated micro-batches of THIS rank
    (``input_ids``, ``labels``, ``attention_mask``, ``audio_encodings``).  Returns the list of logged losses.
    With ``output_dir``: resumes from the newest ``checkpoint-*`` there (m2t/train.py:257-260), saves every
    ``save_steps`` optimizer steps (train_llark.sh:31,41-42) and once more at the end."""
    toks = [t for t in (audio_cfg.audio_start_token, audio_cfg.audio_end_token) if isinstance(t, int)]
    tr = HipLlamaTrainer(engine, lr=cfg.learning_rate, betas=(cfg.adam_beta1, cfg.adam_beta2), eps=cfg.adam_epsilon,
                         weight_decay=cfg.weight_decay, embed_grad_tokens=toks)
    from . import checkpoint as CK

    if output_dir:
        CK.maybe_resume(tr, output_dir)
    losses, acc, micro = [], 0.0, 0
    for batch in batches:
        ids = batch["input_ids"].to(engine.device)
        feats = batch.get("audio_encodings")
        if feats is not None:
            feats = [f.to(engine.device, torch.float32) for f in feats] if isinstance(feats, (list, tuple)) \
                else feats.to(engine.device, torch.float32)
        segs = plan_audio_splice(ids, feats, audio_cfg, False)
        last = (micro + 1) % cfg.gradient_accumulation_steps == 0          # DDP no_sync boundary: exchange only on the last micro-step
        loss = tr.forward_backward(ids, segs, batch["labels"].to(engine.device), 1.0 / cfg.gradient_accumulation_steps,
                                   overlap_allreduce_world=world if last else 1)
        acc += float(loss.item()) / cfg.gradient_accumulation_steps
        micro += 1
        if micro % cfg.gradient_accumulation_steps == 0:
            tr.allreduce_grads(world)                       # the one exchange step of the path
            tr.lr = lr_at(tr.step_count, cfg)
            tr.step(world)
            mean_loss = D.max_over_ranks(acc, 1)            # local value; rank 0 logs
            losses.append(mean_loss)
            if log:
                log(dict(step=tr.step_count, loss=mean_loss, lr=tr.lr))
            acc = 0.0
            if output_dir and tr.step_count % save_steps == 0:
                CK.save_checkpoint(tr, output_dir, save_total_limit, rank)
            if max_optimizer_steps and tr.step_count >= max_optimizer_steps:
                break
    if output_dir:
        CK.save_checkpoint(tr, output_dir, save_total_limit, rank)
    return losses
