"""A directory saved by the reference carries ``"model_type": "wrapped_llamav2"`` (m2t/models/llamav2.py:42,422-423).  When the
reference package is not importable, llark_amd registers that name to the HIP classes, so ``AutoConfig`` /
``AutoModelForCausalLM`` resolve such a directory without the caller naming a class."""
import json
import os

from transformers import AutoConfig, AutoModelForCausalLM


def test_reference_model_type_resolves_to_hip_classes(tmp_path):
    import llark_amd.m2t.llamav2 as L

    assert L.register_reference_names() is True            # /root/reference is not on sys.path in the test suite
    cfg = dict(model_type="wrapped_llamav2", architectures=["WrappedLlamav2ForCausalLM"], hidden_size=64, intermediate_size=128,
               num_hidden_layers=1, num_attention_heads=2, vocab_size=100, mm_hidden_size=48)
    json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    c = AutoConfig.from_pretrained(str(tmp_path))
    assert isinstance(c, L.WrappedLlamav2Config) and c.mm_hidden_size == 48 and c.model_type == "wrapped_llamav2"
    m = AutoModelForCausalLM.from_config(c)
    assert isinstance(m, L.WrappedLlamav2ForCausalLM)
    assert type(m.get_model()).__name__ == "WrappedLlamav2Model" and hasattr(m.get_model(), "initialize_adapter_modules")
    # the library's own name keeps working
    c2 = L.WrappedLlamav2Config(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=100)
    assert c2.model_type == "wrapped_llamav2_hip"
