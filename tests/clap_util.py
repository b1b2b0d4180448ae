"""Shared by the CLAP tests: the inverse of llark_amd.clap.from_laion_state_dict (transformers names -> the names a laion_clap
checkpoint uses, fused qkv), plus junk keys a real checkpoint also carries."""
import torch


def to_laion_names(w):
    e = "audio_model.audio_encoder."
    laion = {"module.text_branch.embeddings.word_embeddings.weight": torch.zeros(2, 2),
             "module.audio_branch.spectrogram_extractor.stft.conv_real.weight": torch.zeros(2, 1, 4),
             "module.audio_branch.layers.0.blocks.0.attn.relative_position_index": torch.zeros(64, 64)}
    inv = ((".layernorm_before.", ".norm1."), (".layernorm_after.", ".norm2."), (".attention.output.dense.", ".attn.proj."),
           (".intermediate.dense.", ".mlp.fc1."), (".output.dense.", ".mlp.fc2."),
           (".attention.self.relative_position_bias_table", ".attn.relative_position_bias_table"))
    for k, v in w.items():
        if k.startswith("audio_projection."):
            laion["module." + k.replace("linear1", "0").replace("linear2", "2")] = v
            continue
        r = k[len(e):]
        if r.startswith("batch_norm."):
            laion["module.audio_branch.bn0." + r[len("batch_norm."):]] = v
            continue
        if ".attention.self.query." in r:
            kk, vv = r.replace(".query.", ".key."), r.replace(".query.", ".value.")
            laion["module.audio_branch." + r.replace(".attention.self.query.", ".attn.qkv.")] = torch.cat([v, w[e + kk], w[e + vv]], 0)
            continue
        if ".attention.self.key." in r or ".attention.self.value." in r:
            continue
        for a, b in inv:
            r = r.replace(a, b)
        laion["module.audio_branch." + r] = v
    return laion
