"""``WrappedMPTForCausalLM`` surface of the reference's ``m2t/models/mpt.py`` on the HIP MPT engine.

The reference class derives from the vendored LLaVA ``MPTForCausalLM`` (a ``transformers.PreTrainedModel``), which no
longer constructs under the installed transformers 5.15 (its config subclass drops its fields).  This
module therefore keeps the *interface* -- parameter names of the state dict (``transformer.wte.weight``,
``transformer.blocks.N.{norm_1,attn.Wqkv,attn.out_proj,norm_2,ffn.up_proj,ffn.down_proj}.*``, ``transformer.norm_f.*``,
``transformer.mm_projector.*``), ``get_model()`` / ``.model``, ``initialize_adapter_modules``,
``initialize_audio_tokenizer``, ``forward(input_ids, ..., labels, audio_encodings)`` -> ``CausalLMOutputWithPast``,
``prepare_inputs_for_generation`` and a greedy ``generate`` -- on plain ``torch.nn.Module`` containers; the arithmetic runs
in ``HipMptEngine``; the training step lives in ``llark_amd.m2t.mpt_train_engine.HipMptTrainer`` (native loop, not wired
into this wrapper's autograd).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch
from torch import nn
from transformers.modeling_outputs import CausalLMOutputWithPast

from .. import ops
from . import AudioEncoderConfig
from .llamav2 import EngineCache, plan_audio_splice
from .mpt_engine import HipMptEngine, MptDims
from .special_tokens import DEFAULT_AUDIO_END_TOKEN, DEFAULT_AUDIO_PATCH_TOKEN, DEFAULT_AUDIO_START_TOKEN

ATTN_DEFAULTS = dict(attn_type="multihead_attention", attn_pdrop=0.0, attn_impl="torch", qk_ln=False, clip_qkv=None, softmax_scale=None,
                     prefix_lm=False, attn_uses_sequence_id=False, alibi=True, alibi_bias_max=8)


@dataclass
class WrappedMPTConfig:
    """Fields of m2t/llava/model/mpt/configuration_mpt.py:MPTConfig that the hot path reads (+ the wrapper's mm fields)."""
    d_model: int = 2048
    n_heads: int = 16
    n_layers: int = 24
    expansion_ratio: int = 4
    max_seq_len: int = 2048
    vocab_size: int = 50368
    attn_config: Dict[str, Any] = field(default_factory=lambda: dict(ATTN_DEFAULTS))
    logit_scale: Any = None
    no_bias: bool = True
    norm_type: str = "low_precision_layernorm"
    use_cache: bool = False
    tie_word_embeddings: bool = True
    use_mm_proj: bool = True
    mm_hidden_size: int = 512
    model_type: str = "wrapped_mpt_hip"

    def validate(self) -> None:
        a = dict(ATTN_DEFAULTS, **self.attn_config)
        self.attn_config = a
        if not self.tie_word_embeddings:
            raise ValueError("MPTForCausalLM only supports tied word embeddings")
        if a["attn_type"] != "multihead_attention" or a["prefix_lm"] or a["attn_uses_sequence_id"] or not a["alibi"]:
            raise NotImplementedError("built: multihead attention, causal, ALiBi (the reference's MPT configurations)")
        if a["softmax_scale"] is not None:
            raise NotImplementedError("softmax_scale overrides are not built (default 1/sqrt(head_dim))")
        if self.norm_type.lower() not in ("low_precision_layernorm", "layernorm"):
            raise NotImplementedError(f"norm_type {self.norm_type!r}")


class _Block(nn.Module):
    def __init__(self, cfg: WrappedMPTConfig):
        super().__init__()
        D, bias = cfg.d_model, not cfg.no_bias
        self.norm_1 = nn.LayerNorm(D, bias=bias)
        self.attn = nn.Module()
        self.attn.Wqkv = nn.Linear(D, 3 * D, bias=bias)
        self.attn.out_proj = nn.Linear(D, D, bias=bias)
        if cfg.attn_config["qk_ln"]:
            self.attn.q_ln, self.attn.k_ln = nn.LayerNorm(D, bias=bias), nn.LayerNorm(D, bias=bias)
        self.norm_2 = nn.LayerNorm(D, bias=bias)
        self.ffn = nn.Module()
        self.ffn.up_proj = nn.Linear(D, cfg.expansion_ratio * D, bias=bias)
        self.ffn.down_proj = nn.Linear(cfg.expansion_ratio * D, D, bias=bias)


class WrappedMPTModel(nn.Module):
    """Parameter container named like the reference's ``transformer`` module; never executed on the CPU."""

    def __init__(self, cfg: WrappedMPTConfig):
        super().__init__()
        self.config = cfg
        self.wte = nn.Embedding(cfg.vocab_size, cfg.d_model)
        self.blocks = nn.ModuleList([_Block(cfg) for _ in range(cfg.n_layers)])
        self.norm_f = nn.LayerNorm(cfg.d_model, bias=not cfg.no_bias)
        self.audio_encoder_config = AudioEncoderConfig()
        if cfg.use_mm_proj:
            self.mm_projector = nn.Linear(cfg.mm_hidden_size, cfg.d_model)

    def initialize_adapter_modules(self, pretrain_mm_mlp_adapter=None, tune_mm_mlp_adapter=False, fsdp=None):
        """m2t/models/mpt.py:47-72."""
        del tune_mm_mlp_adapter, fsdp
        self.config.use_mm_proj = True
        if not hasattr(self, "mm_projector"):
            self.mm_projector = nn.Linear(self.config.mm_hidden_size, self.config.d_model)
        if pretrain_mm_mlp_adapter is not None:
            sd = torch.load(pretrain_mm_mlp_adapter, map_location="cpu")
            self.mm_projector.load_state_dict({k.split(".")[-1]: v for k, v in sd.items() if "mm_projector" in k})
        return dict(audio_config=AudioEncoderConfig())


class WrappedMPTForCausalLM(nn.Module):
    def __init__(self, config: WrappedMPTConfig):
        super().__init__()
        config.validate()
        self.config = config
        self.transformer = WrappedMPTModel(config)
        ls = config.logit_scale
        if isinstance(ls, str):
            if ls != "inv_sqrt_d_model":
                raise ValueError(f"logit_scale={ls!r} is not recognized as an option; use numeric value or 'inv_sqrt_d_model'.")
            ls = 1 / math.sqrt(config.d_model)
        self.logit_scale = ls
        self._engine: Optional[HipMptEngine] = None
        self._engine_max = (8, 512)
        self._engine_precision = "split"

    # ---- reference surface ----------------------------------------------------------------------
    @property
    def model(self):
        return self.transformer

    def get_model(self):
        return self.transformer

    def get_input_embeddings(self):
        return self.transformer.wte

    def get_output_embeddings(self):
        return self.transformer.wte                      # tied

    def resize_token_embeddings(self, n: int) -> None:
        old = self.transformer.wte
        if n == old.num_embeddings:
            return
        new = nn.Embedding(n, old.embedding_dim).to(device=old.weight.device, dtype=old.weight.dtype)
        k = min(n, old.num_embeddings)
        with torch.no_grad():
            new.weight[:k] = old.weight[:k]
        self.transformer.wte = new
        self.config.vocab_size = n
        self._engine = None

    def initialize_audio_tokenizer(self, mm_use_audio_start_end, tokenizer, device, tune_mm_mlp_adapter=False, pretrain_mm_mlp_adapter=None):
        """m2t/models/mpt.py:372-432 (same rules as the Llama wrapper; embeddings are tied, so one table is updated)."""
        ac = self.get_model().audio_encoder_config
        ac.use_audio_start_end = mm_use_audio_start_end
        tokenizer.add_tokens([DEFAULT_AUDIO_PATCH_TOKEN], special_tokens=True)
        self.resize_token_embeddings(len(tokenizer))
        if mm_use_audio_start_end:
            num_new = tokenizer.add_tokens([DEFAULT_AUDIO_START_TOKEN, DEFAULT_AUDIO_END_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
            ac.audio_start_token, ac.audio_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_AUDIO_START_TOKEN, DEFAULT_AUDIO_END_TOKEN])
            if num_new > 0:
                emb = self.get_input_embeddings().weight.data
                emb[-num_new:] = emb[:-num_new].mean(dim=0, keepdim=True)
            if tune_mm_mlp_adapter:
                self.get_model().orig_embeds_params = [self.get_input_embeddings().weight.data.clone().to(device=device)]
            if pretrain_mm_mlp_adapter:
                w = torch.load(pretrain_mm_mlp_adapter, map_location="cpu")["transformer.wte.weight"]
                emb = self.get_input_embeddings().weight.data
                assert num_new == 2
                if emb.shape == w.shape:
                    emb[-num_new:] = w[-num_new:]
                elif w.shape[0] == num_new:
                    emb[-num_new:] = w
                else:
                    raise ValueError(f"Unexpected embed_tokens_weight shape. Pretrained: {w.shape}. Current: {emb.shape}. "
                                     f"Numer of new tokens: {num_new}.")
        ac.audio_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_AUDIO_PATCH_TOKEN])[0]
        self._engine = None

    # ---- engine plumbing ------------------------------------------------------------------------
    def configure_engine(self, max_batch: int = 8, max_seq: int = 512, precision: str = "split") -> None:
        self._engine_max, self._engine_precision, self._engine = (max_batch, max_seq), precision, None

    def sync_engine(self) -> HipMptEngine:
        c, a = self.config, self.config.attn_config
        dev = self.transformer.wte.weight.device
        if dev.type != "cuda":
            raise ops._lib.LlarkHipError("WrappedMPTForCausalLM runs on the HIP engine only: move the model to the GPU (no CPU fallback)")
        dims = MptDims(d_model=c.d_model, n_heads=c.n_heads, n_layers=c.n_layers, expansion_ratio=c.expansion_ratio,
                       vocab_size=self.transformer.wte.num_embeddings, max_seq_len=c.max_seq_len, alibi_bias_max=a["alibi_bias_max"],
                       qk_ln=a["qk_ln"], clip_qkv=a["clip_qkv"], logit_scale=self.logit_scale, mm_hidden_size=c.mm_hidden_size)
        eng = HipMptEngine(dims, dev, *self._engine_max, precision=self._engine_precision)
        eng.load_state_dict(self.state_dict())
        self._engine = eng
        return eng

    @property
    def engine(self) -> HipMptEngine:
        return self._engine if self._engine is not None else self.sync_engine()

    # ---- forward / generate ---------------------------------------------------------------------
    def forward(self, input_ids: torch.LongTensor, past_key_values=None, attention_mask=None, prefix_mask=None, sequence_id=None,
                labels=None, return_dict=None, output_attentions=None, output_hidden_states=None, use_cache=None, audio_encodings=None):
        if prefix_mask is not None or sequence_id is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("prefix_mask / sequence_id / attention or hidden-state outputs are not built")
        if attention_mask is not None and not bool(attention_mask.to(torch.bool).all()):
            raise NotImplementedError("padded batches are not built for the MPT engine (the reference cannot generate with them either)")
        if torch.is_grad_enabled() and labels is not None and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("use llark_amd.m2t.mpt_train_engine.HipMptTrainer for MPT training; this wrapper's forward is inference (torch.no_grad())")
        eng = self.engine
        ids = input_ids.to(eng.device)
        cached = isinstance(past_key_values, EngineCache) and len(past_key_values) > 0
        segs = []
        if audio_encodings is not None and self.config.use_mm_proj and not cached:
            feats = audio_encodings
            feats = [f.to(eng.device, torch.float32) for f in feats] if isinstance(feats, (list, tuple)) else feats.to(eng.device, torch.float32)
            segs = plan_audio_splice(ids, feats, self.transformer.audio_encoder_config, False, patch_branch=True)
        logits = eng.forward_tokens(ids, segs, pos0=eng.cur_len if cached else 0)
        loss = None
        if labels is not None:
            loss = ops.cross_entropy_shifted(logits, labels.to(eng.device))        # roll(-1) + ignore last == shifted CE
        out = CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=EngineCache(eng), hidden_states=None)
        return out if return_dict is not False else (loss, logits, out.past_key_values)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        """m2t/models/mpt.py:339-370."""
        if inputs_embeds is not None:
            raise NotImplementedError("inputs_embeds is not implemented for MPT yet")
        am = kwargs.get("attention_mask")
        am = torch.ones_like(input_ids, dtype=torch.bool) if am is None else am.bool()
        if am[:, -1].sum() != am.shape[0]:
            raise NotImplementedError("MPT does not support generation with right padding.")
        if past_key_values is not None and len(past_key_values) > 0:
            input_ids = input_ids[:, -1].unsqueeze(-1)
        return {"input_ids": input_ids, "attention_mask": am, "prefix_mask": None, "sequence_id": None, "past_key_values": past_key_values,
                "use_cache": kwargs.get("use_cache", True), "audio_encodings": kwargs.get("audio_encodings", None)}

    @torch.no_grad()
    def generate(self, input_ids=None, audio_encodings=None, max_new_tokens: int = 20, stopping_criteria=None, eos_token_id=None, **kwargs):
        """Greedy loop driven by :meth:`prepare_inputs_for_generation` (prompt with audio once, then one token per step)."""
        ids = input_ids.to(self.engine.device)
        past = None
        for _ in range(max_new_tokens):
            inp = self.prepare_inputs_for_generation(ids, past_key_values=past, audio_encodings=audio_encodings)
            out = self.forward(inp["input_ids"], past_key_values=inp["past_key_values"], audio_encodings=inp["audio_encodings"])
            past = out.past_key_values
            scores = out.logits[:, -1]
            nxt = scores.argmax(-1, keepdim=True)
            ids = torch.cat((ids, nxt), dim=1)
            if eos_token_id is not None and bool((nxt == eos_token_id).all()):
                break
            if stopping_criteria is not None and any(bool(c(ids, scores)) for c in stopping_criteria):
                break
        return ids
