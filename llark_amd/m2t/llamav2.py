"""Drop-in for the reference's ``m2t/models/llamav2.py`` running on the MI355X HIP engine.

Same class names, constructor / method signatures, state-dict keys and error behaviour as the
reference (file:line given per method); the arithmetic -- ``embed_tokens`` gather, ``mm_projector``,
the audio splice, the 32 decoder layers, final norm and ``lm_head`` -- runs in
:class:`llark_amd.m2t.engine.HipLlamaEngine` instead of HF's eager PyTorch modules.

Weights stay in the HF ``nn.Parameter`` containers (so ``from_pretrained``, ``state_dict``,
``resize_token_embeddings`` and checkpoint side-files keep working); the engine holds bf16 copies
in kernel layout that are (re)built by :meth:`WrappedLlamav2ForCausalLM.sync_engine` after any
weight change.  Training: ``model(input_ids, labels=..., audio_encodings=...).loss.backward()`` runs the HIP
forward / backward of :class:`llark_amd.m2t.train_engine.HipLlamaTrainer` through an autograd bridge that fills the
``.grad`` of these ``nn.Parameter``s (m2t/train.py:53-277); the RCCL gradient all-reduce lives in ``llark_amd/dist.py``.
"""
from __future__ import annotations

import logging
import os
from typing import List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
from transformers import AutoConfig, AutoModelForCausalLM, LlamaConfig, LlamaForCausalLM, LlamaModel
from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast

from .. import ops
from . import AudioEncoderConfig
from .engine import HipLlamaEngine, LlamaDims
from .special_tokens import DEFAULT_AUDIO_END_TOKEN, DEFAULT_AUDIO_PATCH_TOKEN, DEFAULT_AUDIO_START_TOKEN


class WrappedLlamav2Config(LlamaConfig):
    """m2t/models/llamav2.py:39-43."""

    model_type = "wrapped_llamav2_hip"
    mm_hidden_size: int = 4800  # size of the jukebox embeddings with temporal averaging


def _rope_theta(cfg) -> float:
    """transformers 4.x keeps ``rope_theta`` on the config, 5.x inside ``rope_parameters``."""
    theta = getattr(cfg, "rope_theta", None)
    if theta is None:
        rp = getattr(cfg, "rope_parameters", None) or {}
        theta = rp.get("rope_theta", 10000.0) if isinstance(rp, dict) else 10000.0
    return float(theta)


class EngineCache:
    """Opaque ``past_key_values`` handle: the KV cache lives inside the engine (static, in HBM)."""

    def __init__(self, engine: HipLlamaEngine):
        self.engine = engine

    def get_seq_length(self, *a, **k) -> int:
        return self.engine.cur_len

    def __bool__(self) -> bool:
        return self.engine.cur_len > 0

    def __len__(self) -> int:
        d = self.engine.dims
        return getattr(d, "num_hidden_layers", getattr(d, "n_layers", 1)) if self.engine.cur_len > 0 else 0


def plan_audio_splice(input_ids: torch.Tensor, audio_features: Union[None, torch.Tensor, Sequence[torch.Tensor]],
                      cfg: AudioEncoderConfig, has_past: bool, patch_branch: bool = False) -> List[Tuple[int, int, torch.Tensor]]:
    """Validation + placement rules of m2t/models/llamav2.py:141-222 (use_audio_start_end=True):
    returns [(batch index, position of <audio_start>, frames (F, mm))].  Same ValueErrors.
    ``patch_branch``: the MPT wrapper also implements use_audio_start_end=False (m2t/models/mpt.py:190-232)."""
    if audio_features is None:
        return []
    if not cfg.use_audio_start_end:
        if not patch_branch:
            raise NotImplementedError("audio_encoder_config.use_audio_start_end=False is not implemented.")
        return _plan_patch_splice(input_ids, audio_features, cfg)
    ids = input_ids.detach().cpu()
    segs = []
    cur_audio_idx = 0
    for b in range(ids.shape[0]):
        cur = ids[b]
        n_start = int((cur == cfg.audio_start_token).sum())
        if n_start != int((cur == cfg.audio_end_token).sum()):
            raise ValueError("The number of image start tokens and image end tokens should be the same.")
        starts = torch.where(cur == cfg.audio_start_token)[0]
        if not len(starts) and not has_past:
            logging.warning("no audio start tokens detected and there are no past_key_values;"
                            "if this is a multimodal model this could be a problem.")
        for pos in starts.tolist():
            feats = audio_features[cur_audio_idx]
            num_frames = feats.shape[0]
            if pos + num_frames + 1 >= cur.shape[0] or cur[pos + num_frames + 1] != cfg.audio_end_token:
                raise ValueError("The image end token should follow the image start token.")
            segs.append((b, pos, feats))
            cur_audio_idx += 1
    return segs


def _plan_patch_splice(input_ids, audio_features, cfg: AudioEncoderConfig) -> List[Tuple[int, int, torch.Tensor]]:
    """m2t/models/mpt.py:190-232: exactly num_frames consecutive <audio_patch> tokens per example are replaced by the
    projected frames.  Returned in the (batch, start, frames) convention of the engines (rows start+1 .. start+F)."""
    ids = input_ids.detach().cpu()
    segs = []
    for b in range(ids.shape[0]):
        feats = audio_features[b]
        num_frames = feats.shape[0]
        where = torch.where(ids[b] == cfg.audio_patch_token)[0]
        if len(where) != num_frames:
            raise ValueError("The number of audio patch tokens should be the same as the number of audio frames.")
        first = int(where[0])
        if (where != torch.arange(first, first + num_frames)).any():
            raise ValueError("The image patch tokens should be consecutive.")
        segs.append((b, first - 1, feats))
    return segs


class _HipTrainStep(torch.autograd.Function):
    """``loss = model(input_ids, labels=..., audio_encodings=...)`` followed by ``loss.backward()`` on the HIP training
    kernels: the forward AND the whole backward run eagerly inside ``forward`` (llark_amd.m2t.train_engine); ``backward``
    only hands the per-parameter gradients (reference layouts / dtypes) to autograd, so ``param.grad`` accumulation,
    DDP gradient hooks and any torch optimizer keep working unchanged."""

    @staticmethod
    def forward(ctx, model, input_ids, segs, labels, names, *params):
        tr = model._hip_trainer()
        tr.zero_grad()
        loss = tr.forward_backward(input_ids, segs, labels)
        grads = tr.export_grads_hf()
        ctx.lacking = [n for n in names if n not in grads]
        ctx.grads = [grads.get(n) for n in names]
        ctx.dtypes = [p.dtype for p in params]
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.lacking:  # e.g. lm_head when it is not frozen: a silent None gradient would leave the parameter untrained
            raise NotImplementedError(f"the HIP training step exports no gradient for trainable parameter(s) {ctx.lacking[:4]}; freeze them "
                                      "(requires_grad_(False)) -- the reference recipe trains with lm_head frozen (llamav2.py:395-419)")
        outs = []
        for g, dt in zip(ctx.grads, ctx.dtypes):
            outs.append(None if g is None else (g * grad_out).to(dt))
        return (None, None, None, None, None, *outs)


class WrappedLlamav2Model(LlamaModel):
    """m2t/models/llamav2.py:46-234."""

    config_class = WrappedLlamav2Config

    def __init__(self, config: LlamaConfig):
        super().__init__(config)
        self.audio_encoder_config = AudioEncoderConfig()

    def initialize_adapter_modules(self, pretrain_mm_mlp_adapter=None, tune_mm_mlp_adapter=None, fsdp: bool = None):
        """m2t/models/llamav2.py:60-93: creates ``mm_projector = nn.Linear(mm_hidden_size, hidden_size)``."""
        print("[INFO] ignoring parameter fsdp")
        del fsdp
        self.config.use_mm_proj = True
        if not hasattr(self, "mm_projector"):
            self.mm_projector = nn.Linear(self.config.mm_hidden_size, self.config.hidden_size)
        if pretrain_mm_mlp_adapter is not None:
            mm_projector_weights = torch.load(pretrain_mm_mlp_adapter, map_location="cpu")
            self.mm_projector.load_state_dict(
                {k.split(".")[-1]: v for k, v in mm_projector_weights.items() if "mm_projector" in k})
        return dict(audio_config=AudioEncoderConfig())


class WrappedLlamav2ForCausalLM(LlamaForCausalLM):
    """m2t/models/llamav2.py:237-419 on the HIP engine."""

    config_class = WrappedLlamav2Config
    supports_gradient_checkpointing = True

    def __init__(self, config):
        super(LlamaForCausalLM, self).__init__(config)
        self.model = WrappedLlamav2Model(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()
        self._engine: Optional[HipLlamaEngine] = None
        self._engine_max = (8, 512)
        self._engine_precision = os.environ.get("LLARK_LLM_PRECISION", "split")

    def get_model(self):
        return self.model

    # ---- engine management -------------------------------------------------------------------
    def configure_engine(self, max_batch: int = 8, max_seq: int = 512, precision: Optional[str] = None) -> None:
        """precision: "split" (fp32-class, default; env LLARK_LLM_PRECISION) or "bf16" (reference's GPU dtype flow)."""
        self._engine_max = (max_batch, max_seq)
        if precision is not None:
            self._engine_precision = precision
        self._engine = None

    def sync_engine(self) -> HipLlamaEngine:
        """(Re)builds the kernel-layout weight copies from the module's parameters."""
        cfg = self.config
        dev = self.lm_head.weight.device
        if dev.type != "cuda":
            raise ops._lib.LlarkHipError("WrappedLlamav2ForCausalLM runs on the GPU only (there is no CPU fallback); "
                                         "call .cuda() first")
        dims = LlamaDims(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                         num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                         vocab_size=self.lm_head.weight.shape[0], rms_norm_eps=cfg.rms_norm_eps,
                         rope_theta=_rope_theta(cfg),
                         mm_hidden_size=getattr(cfg, "mm_hidden_size", 4800))
        if getattr(cfg, "num_key_value_heads", cfg.num_attention_heads) != cfg.num_attention_heads:
            raise NotImplementedError("grouped-query attention is not used by Llama-2-7B and is not built")
        eng = HipLlamaEngine(dims, dev, *self._engine_max, precision=self._engine_precision)
        sd = {k: v for k, v in self.state_dict().items()}
        eng.load_state_dict(sd)
        self._engine = eng
        return eng

    @property
    def engine(self) -> HipLlamaEngine:
        return self._engine if self._engine is not None else self.sync_engine()

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, input_ids: torch.LongTensor = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.Tensor] = None, past_key_values=None,
                labels: Optional[torch.LongTensor] = None, use_cache: Optional[bool] = None,
                output_attentions: Optional[bool] = None, output_hidden_states: Optional[bool] = None,
                return_dict: Optional[bool] = None, audio_encodings=None, **kwargs):
        """m2t/models/llamav2.py:259-337.  Returns CausalLMOutputWithPast(loss, logits, past_key_values)."""
        if torch.is_grad_enabled() and labels is not None and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(input_ids, labels, audio_encodings, attention_mask, return_dict)
        if output_attentions:
            raise NotImplementedError("output_attentions: the flash-style attention kernels never materialise the S x S probabilities "
                                      "(unused by m2t/; m2t/models/llamav2.py:259-270 only forwards the flag)")
        if position_ids is not None:
            raise NotImplementedError("custom position_ids are not used by the reference path")
        return_dict = True if return_dict is None else return_dict
        eng = self.engine
        input_ids = input_ids.to(eng.device)
        B, S = input_ids.shape
        if attention_mask is not None:
            am = attention_mask.to(torch.bool)
            total = am.shape[1]
            # right padding + causal attention never lets a valid position see a pad: masks of the
            # form 1..1 0..0 (the reference's collator, m2t/data_modules.py:189-222) need no kernel support
            ok = bool((am[:, :-1] | ~am[:, 1:]).all()) if total > 1 else True
            if not ok:
                raise NotImplementedError("only right-padded attention masks are supported")
        has_past = bool(past_key_values) if past_key_values is not None else False
        pos0 = eng.cur_len if has_past else 0
        cfg = self.model.audio_encoder_config
        feats = audio_encodings
        if feats is not None and getattr(self.config, "use_mm_proj", False):
            if isinstance(feats, (list, tuple)):
                feats = [f.to(device=eng.device, dtype=torch.float32) for f in feats]
            else:
                feats = feats.to(device=eng.device, dtype=torch.float32)
            segs = plan_audio_splice(input_ids, feats, cfg, has_past)
        else:
            segs = []
        hidden = [] if output_hidden_states else None      # HF layout: the stream entering every layer, then norm(last) -- (L + 1) x (B, S, H)
        logits = eng.forward_tokens(input_ids, segs, pos0=pos0, hidden_sink=hidden)
        loss = None
        if labels is not None:
            loss = ops.cross_entropy_shifted(logits, labels.to(eng.device))
        cache = EngineCache(eng)
        hs = tuple(hidden) if hidden is not None else None
        if not return_dict:
            out = (logits, cache) + ((hs,) if hs is not None else ())
            return (loss,) + out if loss is not None else out
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=cache, hidden_states=hs, attentions=None)

    # ---- training (m2t/train.py path) ---------------------------------------------------------
    def _hip_trainer(self):
        from .train_engine import HipLlamaTrainer

        if getattr(self, "_trainer", None) is None or self._trainer.eng is not self._train_engine:
            cfg = self.model.audio_encoder_config
            toks = [t for t in (cfg.audio_start_token, cfg.audio_end_token) if isinstance(t, int)]
            orig = getattr(self.model, "orig_embeds_params", None)
            # the torch optimizer of the caller owns the AdamW state on this path: no moments here
            self._trainer = HipLlamaTrainer(self._train_engine, embed_grad_tokens=toks, train_embed_all=orig is None,
                                            optimizer_state=False)
        # HF: model.gradient_checkpointing_enable() flips `model.model.gradient_checkpointing` (supports_gradient_checkpointing = True)
        self._trainer.gradient_checkpointing = bool(getattr(self.model, "gradient_checkpointing", False))
        return self._trainer

    def _forward_train(self, input_ids, labels, audio_encodings, attention_mask, return_dict):
        """Training forward (bf16 flow, like the reference's bf16 recipe).  The kernel-layout weights are refreshed from the
        nn.Parameters on every call because an external optimizer may have changed them -- IN PLACE into one cached training
        engine (and one cached trainer: its flat gradient buffer is allocated once, not per step)."""
        if attention_mask is not None:
            am = attention_mask.to(torch.bool)
            if am.shape[1] > 1 and not bool((am[:, :-1] | ~am[:, 1:]).all()):
                raise NotImplementedError("only right-padded attention masks are supported")
        eng = getattr(self, "_train_engine", None)
        shape_key = (self.lm_head.weight.shape[0], self.lm_head.weight.device)
        if eng is None or getattr(self, "_train_engine_key", None) != shape_key:
            prec = self._engine_precision
            self._engine_precision = "bf16"
            try:
                self._train_engine = self.sync_engine()
            finally:
                self._engine_precision = prec
            self._train_engine_key = shape_key
            self._trainer = None
            eng = self._train_engine
        else:
            from .checkpoint import copy_weights_into_engine

            copy_weights_into_engine(eng, self.state_dict(), origin="module parameters")
        self._engine = None                                     # the inference engine is rebuilt lazily in its own mode
        input_ids = input_ids.to(eng.device)
        feats = audio_encodings
        segs = []
        if feats is not None and getattr(self.config, "use_mm_proj", False):
            feats = [f.to(device=eng.device, dtype=torch.float32) for f in feats] if isinstance(feats, (list, tuple)) \
                else feats.to(device=eng.device, dtype=torch.float32)
            segs = plan_audio_splice(input_ids, feats, self.model.audio_encoder_config, False)
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        names = [n for n, _ in named]
        loss = _HipTrainStep.apply(self, input_ids, segs, labels.to(eng.device), names, *[p for _, p in named])
        if return_dict is False:
            return (loss, None)
        return CausalLMOutputWithPast(loss=loss, logits=None, past_key_values=None, hidden_states=None, attentions=None)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None,
                                      **kwargs):
        """m2t/models/llamav2.py:339-365."""
        if past_key_values:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            raise NotImplementedError("inputs_embeds are not accepted by the HIP engine")
        model_inputs = {"input_ids": input_ids}
        model_inputs.update({
            "attention_mask": attention_mask,
            "past_key_values": past_key_values,
            "use_cache": kwargs.get("use_cache", True),
            "audio_encodings": kwargs.get("audio_encodings", None),
        })
        return model_inputs

    @torch.no_grad()
    def generate(self, input_ids=None, audio_encodings=None, max_new_tokens: int = 20, do_sample: bool = False,
                 stopping_criteria=None, eos_token_id=None, temperature: float = 1.0, **kwargs):
        """Greedy / temperature sampling loop with the semantics of HF 4.29.2 ``generate`` driven by
        :meth:`prepare_inputs_for_generation`: the full prompt (with audio) once, then one token per step
        against the engine's KV cache; ``stopping_criteria`` are called as ``c(ids, scores)`` each step
        (m2t/generate.py:31-44, m2t/infer.py:146-152)."""
        ids = input_ids.to(self.engine.device)
        eos = eos_token_id if eos_token_id is not None else getattr(self.generation_config, "eos_token_id", None)
        pad = getattr(self.generation_config, "pad_token_id", None)
        pad = eos if pad is None else pad
        unfinished = torch.ones((ids.shape[0], 1), dtype=torch.bool, device=ids.device)      # HF: unfinished_sequences, per row
        past = None
        for _ in range(max_new_tokens):
            inputs = self.prepare_inputs_for_generation(ids, past_key_values=past, audio_encodings=audio_encodings)
            eng = self.engine
            step_ids = inputs["input_ids"]
            if past:
                logits = eng.forward_tokens(step_ids, (), pos0=eng.cur_len, last_only=True)
            else:
                feats = inputs["audio_encodings"]
                segs = []
                if feats is not None:
                    if isinstance(feats, (list, tuple)):
                        feats = [f.to(device=eng.device, dtype=torch.float32) for f in feats]
                    else:
                        feats = feats.to(device=eng.device, dtype=torch.float32)
                    segs = plan_audio_splice(step_ids, feats, self.model.audio_encoder_config, False)
                logits = eng.forward_tokens(step_ids, segs, pos0=0, last_only=True)
            past = EngineCache(eng)
            scores = logits[:, -1]
            if do_sample:
                probs = torch.softmax(scores / max(temperature, 1e-6), dim=-1)
                nxt = torch.multinomial(probs, 1)
            else:
                nxt = scores.argmax(-1, keepdim=True)
            if eos is not None and eos >= 0:
                nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))                 # finished rows emit padding
                unfinished = unfinished & (nxt != eos)
            ids = torch.cat((ids, nxt), dim=1)
            if eos is not None and eos >= 0 and not bool(unfinished.any()):
                break
            if stopping_criteria is not None and any(bool(c(ids, scores)) for c in stopping_criteria):
                break
        return ids

    def initialize_audio_tokenizer(self, mm_use_audio_start_end, tokenizer, device, tune_mm_mlp_adapter=False,
                                   pretrain_mm_mlp_adapter=None):
        """m2t/models/llamav2.py:367-419: adds <audio_patch>/<audio_start>/<audio_end>, resizes the
        embeddings, mean-initialises the new rows, records ``orig_embeds_params``, freezes lm_head."""
        del pretrain_mm_mlp_adapter
        audio_encoder_config = self.get_model().audio_encoder_config
        audio_encoder_config.use_audio_start_end = mm_use_audio_start_end
        tokenizer.add_tokens([DEFAULT_AUDIO_PATCH_TOKEN], special_tokens=True)
        self.resize_token_embeddings(len(tokenizer))
        if mm_use_audio_start_end:
            num_new_tokens = tokenizer.add_tokens([DEFAULT_AUDIO_START_TOKEN, DEFAULT_AUDIO_END_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
            audio_encoder_config.audio_start_token, audio_encoder_config.audio_end_token = tokenizer.convert_tokens_to_ids(
                [DEFAULT_AUDIO_START_TOKEN, DEFAULT_AUDIO_END_TOKEN])
            if num_new_tokens > 0:
                input_embeddings = self.get_input_embeddings().weight.data
                output_embeddings = self.get_output_embeddings().weight.data
                input_embeddings[-num_new_tokens:] = input_embeddings[:-num_new_tokens].mean(dim=0, keepdim=True)
                output_embeddings[-num_new_tokens:] = output_embeddings[:-num_new_tokens].mean(dim=0, keepdim=True)
            if tune_mm_mlp_adapter:
                self.get_model().orig_embeds_params = [self.get_input_embeddings().weight.data.clone().to(device=device)]
                for p in self.get_input_embeddings().parameters():
                    p.requires_grad = True
                for p in self.get_output_embeddings().parameters():
                    p.requires_grad = False
        audio_encoder_config.audio_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_AUDIO_PATCH_TOKEN])[0]
        self._engine = None      # vocabulary changed: rebuild the kernel-layout copies lazily


try:
    AutoConfig.register("wrapped_llamav2_hip", WrappedLlamav2Config)
    AutoModelForCausalLM.register(WrappedLlamav2Config, WrappedLlamav2ForCausalLM)
except ValueError:      # already registered (module re-import)
    pass


class WrappedLlamav2ReferenceConfig(WrappedLlamav2Config):
    """The reference's own ``model_type`` (m2t/models/llamav2.py:42,422): a directory saved by the reference
    (``config.json`` carrying "wrapped_llamav2") resolves to the HIP classes through ``AutoConfig`` /
    ``AutoModelForCausalLM`` without the caller naming them."""

    model_type = "wrapped_llamav2"


class WrappedLlamav2ReferenceForCausalLM(WrappedLlamav2ForCausalLM):
    config_class = WrappedLlamav2ReferenceConfig


def register_reference_names(force: bool = False) -> bool:
    """Registers the alias above -- only where it cannot collide: when the reference package is not importable (its module
    registers the same name at import and ``AutoConfig.register`` refuses duplicates), or when forced
    (``LLARK_REGISTER_REFERENCE_NAMES=1`` / ``force=True``).  Returns whether the alias is registered to the HIP classes."""
    import importlib.util

    if not force and os.environ.get("LLARK_REGISTER_REFERENCE_NAMES", "") != "1":
        try:
            if importlib.util.find_spec("m2t.models.llamav2") is not None:
                return False
        except (ImportError, ValueError):
            pass
    try:
        AutoConfig.register("wrapped_llamav2", WrappedLlamav2ReferenceConfig)
        AutoModelForCausalLM.register(WrappedLlamav2ReferenceConfig, WrappedLlamav2ReferenceForCausalLM)
    except ValueError:
        from transformers.models.auto.configuration_auto import CONFIG_MAPPING

        try:
            return CONFIG_MAPPING["wrapped_llamav2"] is WrappedLlamav2ReferenceConfig
        except KeyError:
            return False
    return True


register_reference_names()
