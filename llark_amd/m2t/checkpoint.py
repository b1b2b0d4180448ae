"""Checkpoint / resume of the native training loop, keeping the reference's on-disk contracts:

* ``<output_dir>/checkpoint-<step>/`` holds the full model under the reference's state-dict names
  (``model.layers.N.self_attn.q_proj.weight`` ... ``model.mm_projector.{weight,bias}``, ``lm_head.weight``), so
  ``m2t/models/utils.py:load_pretrained_model`` / ``load_sharded_checkpoint`` and this package's
  ``WrappedLlamav2ForCausalLM.load_state_dict`` read it unchanged;
* ``<output_dir>/mm_projector/checkpoint-<step>.bin`` is the adapter side-file of ``m2t/models/trainer.py:35-65``:
  every key containing ``mm_projector`` / ``embed_tokens`` / ``embed_in`` (``torch.save`` of a plain dict);
* ``save_total_limit`` (``train_llark.sh:42``) prunes old ``checkpoint-*`` folders; training resumes from the newest one
  when any exists (``m2t/train.py:257-260``).

Added for exact resumption (the reference relies on HF Trainer's optimizer.pt / scheduler.pt / trainer_state.json): the
fp32 AdamW moments and the step counter in ``trainer_state.pt`` (kernel layout, one flat tensor each).
"""
from __future__ import annotations

import glob
import os
import re
import shutil
from typing import Dict, Optional

import torch

ADAPTER_KEYS = ("mm_projector", "embed_tokens", "embed_in")
_STEP = re.compile(r"checkpoint-(\d+)$")


def _step_of(path: str) -> int:
    m = _STEP.search(path.rstrip("/"))
    return int(m.group(1)) if m else -1


def list_checkpoints(output_dir: str):
    """``checkpoint-*`` folders, oldest first."""
    return sorted((p for p in glob.glob(os.path.join(output_dir, "checkpoint-*")) if os.path.isdir(p) and _step_of(p) >= 0),
                  key=_step_of)


def latest_checkpoint(output_dir: str) -> Optional[str]:
    cks = list_checkpoints(output_dir)
    return cks[-1] if cks else None


def adapter_state(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in state_dict.items() if any(m in k for m in ADAPTER_KEYS)}


def save_checkpoint(trainer, output_dir: str, save_total_limit: Optional[int] = 1, rank: int = 0, hf_config=None,
                    tokenizer=None) -> Optional[str]:
    """Writes ``checkpoint-<trainer.step_count>`` (+ the adapter side-file) from rank 0; other ranks return None.

    ``hf_config`` / ``tokenizer``: what HF ``Trainer._save`` puts next to the weights and what the inference entry
    needs to open the folder -- ``m2t/models/utils.py:132-172`` (``load_pretrained_model``) calls
    ``AutoTokenizer.from_pretrained(ckpt_dir)`` and ``WrappedLlamav2ForCausalLM.from_pretrained(ckpt_dir)``: ``config.json``
    (resized ``vocab_size``, ``mm_hidden_size``, ``tune_mm_mlp_adapter``, ``mm_use_audio_start_end``) and the tokenizer
    files with the added ``[PAD]`` / audio tokens."""
    if rank != 0:
        return None
    step = trainer.step_count
    folder = os.path.join(output_dir, f"checkpoint-{step}")
    tmp = folder + ".tmp"
    os.makedirs(tmp, exist_ok=True)
    sd = {k: v.detach().cpu().contiguous() for k, v in trainer.eng.state_dict_hf().items()}
    torch.save(sd, os.path.join(tmp, "pytorch_model.bin"))
    torch.save({"step": step, "lr": trainer.lr, "betas": trainer.betas, "eps": trainer.eps, "weight_decay": trainer.wd,
                "exp_avg": trainer.flat_m.cpu(), "exp_avg_sq": trainer.flat_v.cpu(),
                "param_order": [n for n, _ in trainer.params]}, os.path.join(tmp, "trainer_state.pt"))
    if hf_config is not None:
        hf_config.vocab_size = int(sd["model.embed_tokens.weight"].shape[0])
        hf_config.save_pretrained(tmp)
    if tokenizer is not None:
        tokenizer.save_pretrained(tmp)
    if os.path.isdir(folder):
        shutil.rmtree(folder)
    os.replace(tmp, folder)                                   # a crash never leaves a half-written checkpoint-N
    side = os.path.join(output_dir, "mm_projector")
    os.makedirs(side, exist_ok=True)
    torch.save(adapter_state(sd), os.path.join(side, f"checkpoint-{step}.bin"))
    if save_total_limit:
        for old in list_checkpoints(output_dir)[:-save_total_limit]:
            shutil.rmtree(old)
    return folder


def copy_weights_into_engine(eng, sd: Dict[str, torch.Tensor], origin: str = "state dict") -> None:
    """Copies HF-named weights into the engine's kernel-layout tensors IN PLACE (views from ``state_dict_hf``), so that
    whatever points at those tensors (a trainer's parameter table, recorded launch lists) stays valid."""
    cur = eng.state_dict_hf()
    missing = [k for k in cur if k not in sd]
    if missing:
        raise KeyError(f"{origin}: lacks {missing[:3]}{'...' if len(missing) > 3 else ''}")
    for k, dst in cur.items():                                # views into the kernel-layout tensors: copy in place so that
        src = sd[k].to(device=dst.device, dtype=dst.dtype)    # the trainer's parameter table keeps pointing at them
        if dst.shape != src.shape:
            raise ValueError(f"{origin}: {k} has shape {tuple(src.shape)}, engine expects {tuple(dst.shape)}")
        if k.endswith("gate_proj.weight") or k.endswith("up_proj.weight"):
            continue                                          # interleaved storage: handled below
        dst.copy_(src)
    d = eng.dims
    for i, L in enumerate(eng.layers):
        gate, up = sd[f"model.layers.{i}.mlp.gate_proj.weight"], sd[f"model.layers.{i}.mlp.up_proj.weight"]
        gu = L.wgu.view(d.intermediate_size // 32, 2, 32, d.hidden_size)
        gu[:, 0].copy_(gate.to(gu.device, gu.dtype).view(-1, 32, d.hidden_size))
        gu[:, 1].copy_(up.to(gu.device, gu.dtype).view(-1, 32, d.hidden_size))


def load_checkpoint(trainer, folder: str) -> int:
    """Restores weights (kernel layout, in place), AdamW moments and the step counter; returns the step."""
    sd = torch.load(os.path.join(folder, "pytorch_model.bin"), map_location="cpu")
    copy_weights_into_engine(trainer.eng, sd, origin=folder)
    st = torch.load(os.path.join(folder, "trainer_state.pt"), map_location="cpu")
    if st["param_order"] != [n for n, _ in trainer.params]:
        raise ValueError(f"{folder}: optimizer state was written for a different parameter table")
    trainer.flat_m.copy_(st["exp_avg"])
    trainer.flat_v.copy_(st["exp_avg_sq"])
    trainer.step_count = int(st["step"])
    trainer.zero_grad()
    return trainer.step_count


def maybe_resume(trainer, output_dir: str) -> int:
    """m2t/train.py:257-260: resume from the newest ``checkpoint-*`` if there is one; returns the step (0 = fresh)."""
    ck = latest_checkpoint(output_dir) if os.path.isdir(output_dir) else None
    return load_checkpoint(trainer, ck) if ck else 0
