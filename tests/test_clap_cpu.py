"""CPU: host-side logic of the CLAP plugin (no compute calls): the bicubic tap table is ATen's, the laion checkpoint rename
produces exactly the names the engine and the oracle use."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from llark_amd.clap.htsat import ClapDims, bicubic_time_taps, from_laion_state_dict
from oracle import clap_ref as CR


@pytest.mark.parametrize("frames,out", [(1001, 1024), (37, 64), (5, 16), (1024, 1024), (1, 8)])
def test_bicubic_taps_match_aten(frames, out):
    idx, w = bicubic_time_taps(frames, out)
    x = torch.randn(2, 1, frames, 3, generator=torch.Generator().manual_seed(frames))
    ref = F.interpolate(x, (out, 3), mode="bicubic", align_corners=True) if frames < out else x
    got = sum(torch.from_numpy(w[:, j])[None, None, :, None] * x[:, :, torch.from_numpy(idx[:, j]).long(), :] for j in range(4))
    assert (got - ref).abs().max().item() <= 4e-6
    assert np.allclose(w.sum(1), 1.0, atol=1e-6)


def test_too_long_input_rejected_like_the_reference_model():
    with pytest.raises(ValueError, match="less than or equal to the swin input size"):
        bicubic_time_taps(1025, 1024)


def test_laion_checkpoint_rename_round_trip():
    spec = CR.ClapSpec(embed_dim=32, depths=[1, 1], heads=[1, 2], proj_dim=16)
    w = CR.make_weights(spec, seed=1)
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
    back = from_laion_state_dict(laion)
    assert sorted(back) == sorted(w)
    assert all(torch.equal(back[k], w[k]) for k in w)
    assert ClapDims().out_width == 1024
