"""CPU: host-side logic of the CLAP plugin (no compute calls): the bicubic tap table is ATen's, the laion checkpoint rename
produces exactly the names the engine and the oracle use."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from llark_amd.clap.htsat import ClapDims, bicubic_time_taps, from_laion_state_dict
from clap_util import to_laion_names
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
    laion = to_laion_names(w)
    back = from_laion_state_dict(laion)
    assert sorted(back) == sorted(w)
    assert all(torch.equal(back[k], w[k]) for k in w)
    assert ClapDims().out_width == 1024


def test_front_end_host_logic_matches_oracle_and_published_filters():
    """Slaney filter bank == the one transformers.ClapFeatureExtractor publishes (golden) == the oracle's; rand_trunc /
    repeatpad and the int16 round trip of load_audio_input == the oracle's restatement."""
    import os
    from llark_amd.clap import fit_clip, load_audio_input, slaney_mel_filters
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clap_mel.npz"))
    bank = slaney_mel_filters()
    assert bank.shape == (64, 513) and np.abs(bank.T - z["filters"]).max() < 1e-8
    assert np.abs(bank - CR.mel_filterbank_slaney()).max() < 1e-8
    rng = np.random.default_rng(0)
    for n in (7, 16, 17, 40):
        w = rng.standard_normal(n).astype(np.float32)
        off = int(np.random.default_rng(5).integers(0, max(n - 17, 0) + 1)) if n > 17 else 0
        assert np.array_equal(fit_clip(w, np.random.default_rng(5), max_len=17), CR.fit_length(w, 17, off))
    w = (rng.standard_normal(1000) * 0.7).astype(np.float32)
    el = load_audio_input({"waveform": w})
    feats = el["audio_features"][0]["waveform"].numpy()
    assert feats.shape == (480000,) and np.array_equal(feats, CR.fit_length(CR.quantize_roundtrip(w)))
    with pytest.raises(ValueError):
        fit_clip(np.zeros(0, np.float32))


def test_oracle_logmel_pinned_against_feature_extractor():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clap_mel.npz"))
    lm = CR.logmel(CR.fit_length(z["wave"]))
    assert lm.shape == (1001, 64) and np.abs(lm - z["logmel"]).max() < 2e-5
    assert lm.min() == -100.0                                   # the zero tail of repeatpad clamps at amin


def test_embed_cli_host_side(tmp_path):
    """File listing, rank sharding, wav decoding (int16 / float / stereo / other rates) and batching; no GPU involved."""
    from scipy.io import wavfile
    from llark_amd.clap.embed_cli import iter_batches, list_wavs, read_wav_48k, shard
    rng = np.random.default_rng(0)
    (tmp_path / "sub").mkdir()
    a = (rng.standard_normal(4800) * 3000).astype(np.int16)
    wavfile.write(tmp_path / "a.wav", 48000, a)
    wavfile.write(tmp_path / "sub" / "b.WAV", 48000, rng.standard_normal((2400, 2)).astype(np.float32) * 0.1)
    wavfile.write(tmp_path / "c.wav", 44100, (rng.standard_normal(4410) * 0.1).astype(np.float32))
    (tmp_path / "broken.wav").write_bytes(b"not a wav")
    (tmp_path / "notes.txt").write_text("x")
    paths = list_wavs(str(tmp_path))
    assert [p.split("/")[-1] for p in paths] == ["a.wav", "broken.wav", "c.wav", "b.WAV"]
    assert shard(paths, 0, 2) + shard(paths, 1, 2) != [] and sorted(shard(paths, 0, 2) + shard(paths, 1, 2)) == sorted(paths)
    xa = read_wav_48k(paths[0])
    assert xa.dtype == np.float32 and np.array_equal(xa, a.astype(np.float32) / 32768.0)
    assert read_wav_48k(paths[3]).shape == (2400,) and read_wav_48k(paths[2]).shape == (4800,)
    got = list(iter_batches(paths, 2))
    assert [len(n) for n, _ in got] == [2, 1] and got[0][0] == [paths[0], paths[2]]          # broken.wav skipped


def test_synthetic_weights_and_flop_accounting_match_the_oracle_model():
    """llark_amd.clap.random_state_dict (bench / smoke weights) has exactly the oracle's parameter names and shapes, and the
    algorithmic FLOP count bench.py's roofline uses equals a direct count over those shapes."""
    from llark_amd.clap import ClapDims, algorithmic_flops_per_clip, random_state_dict
    for kw in (dict(embed_dim=32, depths=[2, 2, 2, 1], heads=[1, 2, 4, 8], proj_dim=64), {}):
        sd = random_state_dict(ClapDims(**kw), seed=1)
        ref = CR.make_weights(CR.ClapSpec(**kw), seed=1)
        assert sorted(sd) == sorted(ref) and all(sd[k].shape == ref[k].shape for k in ref)
        assert all(torch.isfinite(v).all() for v in sd.values()) and (sd["audio_model.audio_encoder.batch_norm.running_var"] > 0).all()
        d = ClapDims(**kw)
        L, gemm = (d.spec_size // d.patch) ** 2, 0.0
        tokens = {}
        for s in range(len(d.depths)):
            tokens[s] = L // 4 ** s
        for k, v in ref.items():
            if not k.endswith(".weight") or v.dim() < 2:
                continue
            if ".layers." in k:
                s = int(k.split(".layers.")[1].split(".")[0])
                rows = tokens[s] // 4 if ".downsample." in k else tokens[s]
            elif "patch_embed.proj" in k:
                rows, v = L, v.reshape(v.shape[0], -1)
            else:
                rows = 1                                       # projection head: one pooled row per clip
            gemm += 2.0 * rows * v.shape[0] * v.shape[1]
        fl = algorithmic_flops_per_clip(d)
        assert abs(fl["gemm"] - gemm) <= 1e-9 * gemm, (fl["gemm"], gemm)
    assert abs(algorithmic_flops_per_clip()["gemm"] / 1e9 - 29.81) < 0.01
