"""Checkpoint ingestion of ``load_model``: upstream ``.pth.tar`` format, DDP ``module.`` prefix,
``strict=False`` layer dropping (jukebox/main.py:176-200, jukebox/make_models.py.patch:7-8).  CPU: file format and
name filtering only; tests/test_extract_gpu.py loads such files through the HIP path."""
import io
import os

import pytest
import torch

from llark_amd.jukebox import checkpoint as CK
from llark_amd.jukebox.hparams import hparams_tiny
from llark_amd.jukebox.synthetic import make_jukebox_weights


def upstream_style_checkpoints(hps, ckpt_depth, seed=0, prefix="module.", base=None):
    """What upstream's two files hold for a model of this shape: the VQ-VAE file carries ALL levels (encoders, decoders,
    three codebooks + their EMA statistics), the prior file carries ``ckpt_depth`` layers (72 for 5b) plus the
    tensors the only_encode path never touches (x_out, start_token ...), keys under DDP's ``module.`` prefix.
    ``base``: weights of the model under test (its layers come first; the extra layers are copies of other seeds)."""
    w = make_jukebox_weights(hps, seed=seed, depth=ckpt_depth)
    if base is not None:
        w = {**w, **{k: v.clone() for k, v in base.items()}}
    vq = {k: v for k, v in w.items() if k.startswith(("encoders.", "bottleneck."))}
    g = torch.Generator().manual_seed(99)
    for lvl in (0, 1):
        vq[f"encoders.{lvl}.level_blocks.0.model.0.0.weight"] = torch.randn(4, 1, 4, generator=g)
        vq[f"decoders.{lvl}.level_blocks.0.model.0.weight"] = torch.randn(4, 4, 3, generator=g)
        vq[f"bottleneck.level_blocks.{lvl}.k"] = torch.randn(hps.l_bins, hps.emb_width, generator=g)
    vq["bottleneck.level_blocks.2.k_sum"] = torch.randn(hps.l_bins, hps.emb_width, generator=g)
    vq["bottleneck.level_blocks.2.k_elem"] = torch.ones(hps.l_bins)
    vq["decoders.2.out.weight"] = torch.randn(1, 4, 3, generator=g)
    pr = {k: v for k, v in w.items() if k.startswith(("prior.", "y_emb."))}
    pr["prior.x_out.weight"] = torch.randn(hps.l_bins, hps.prior_width, generator=g)
    pr["prior.start_token"] = torch.randn(1, hps.prior_width, generator=g)
    return ({"model": {prefix + k: v for k, v in vq.items()}, "step": 12345, "hps": {"sr": 44100}},
            {"model": {prefix + k: v for k, v in pr.items()}, "step": 777}, w)


def test_pth_tar_roundtrip_drops_layers_beyond_depth(tmp_path):
    hps = hparams_tiny()
    depth = hps.prior_depth
    vq_ck, pr_ck, w = upstream_style_checkpoints(hps, ckpt_depth=2 * depth)       # 5b: 72 layers on disk, 36 built
    pv, pp = tmp_path / "vqvae.pth.tar", tmp_path / "prior_level_2.pth.tar"
    torch.save(vq_ck, pv)
    torch.save(pr_ck, pp)
    got, unexpected = CK.load_checkpoint_weights("5b", hps, depth, str(pv), str(pp))
    assert sorted(got) == sorted(CK.vqvae_names(hps) + CK.prior_names(hps, depth))
    for k, v in got.items():
        assert torch.equal(v, w[k]), k
    dropped = {int(k.split(".")[3]) for k in unexpected if k.startswith("prior.transformer._attn_mods.")}
    assert dropped == set(range(depth, 2 * depth))
    for k in ("prior.x_out.weight", "prior.start_token", "decoders.2.out.weight", "bottleneck.level_blocks.0.k",
              "bottleneck.level_blocks.2.k_sum"):
        assert k in unexpected
    assert not any(k.startswith("module.") for k in unexpected)


def test_file_objects_and_plain_keys():
    hps = hparams_tiny()
    vq_ck, pr_ck, w = upstream_style_checkpoints(hps, ckpt_depth=hps.prior_depth, prefix="")
    buf = io.BytesIO()
    torch.save(pr_ck, buf)
    buf.seek(0)
    sd = CK.read_pth_tar(buf)
    assert torch.equal(sd["prior.x_emb.weight"], w["prior.x_emb.weight"])


def test_missing_tensor_is_an_error_not_a_silent_random_init(tmp_path):
    hps = hparams_tiny()
    vq_ck, pr_ck, _ = upstream_style_checkpoints(hps, ckpt_depth=hps.prior_depth - 1)     # one layer short
    with pytest.raises(KeyError, match=r"_attn_mods"):
        CK.select_weights([CK.read_pth_tar(_save(vq_ck)), CK.read_pth_tar(_save(pr_ck))], hps)
    with pytest.raises(ValueError, match="not a Jukebox checkpoint"):
        CK.read_pth_tar(_save({"state_dict": {}}))


def _save(obj):
    buf = io.BytesIO()
    torch.save(obj, buf)
    buf.seek(0)
    return buf


def test_default_paths_and_absent_files(tmp_path, monkeypatch):
    monkeypatch.setenv(CK.CACHE_ENV, str(tmp_path))
    pv, pp = CK.default_checkpoint_paths("5b_lyrics")
    assert pv == os.path.join(str(tmp_path), "jukebox", "models", "5b", "vqvae.pth.tar")
    assert pp == os.path.join(str(tmp_path), "jukebox", "models", "5b_lyrics", "prior_level_2.pth.tar")
    with pytest.raises(FileNotFoundError, match="vqvae.pth.tar"):
        CK.load_checkpoint_weights("5b", hparams_tiny())
    with pytest.raises(ValueError):
        CK.default_checkpoint_paths("1b_lyrics")


class _Unsafe:                       # a global the weights_only unpickler refuses (stands in for arbitrary pickled code)
    def __init__(self):
        self.x = 1


def test_unsafe_pickle_is_opt_in(tmp_path, monkeypatch):
    """a file the restricted loader rejects is not silently retried with the full unpickler."""
    hps = hparams_tiny()
    _, pr_ck, w = upstream_style_checkpoints(hps, ckpt_depth=hps.prior_depth, prefix="")
    pr_ck = dict(pr_ck, hps=_Unsafe())
    with pytest.raises(ValueError, match="allow_pickle"):
        CK.read_pth_tar(_save(pr_ck))
    sd = CK.read_pth_tar(_save(pr_ck), allow_pickle=True)
    assert torch.equal(sd["prior.x_emb.weight"], w["prior.x_emb.weight"])
    monkeypatch.setenv(CK.UNSAFE_PICKLE_ENV, "1")
    assert "prior.x_emb.weight" in CK.read_pth_tar(_save(pr_ck))


def test_legacy_torch14_serialisation_with_fp16_conv_weights(tmp_path):
    """The released 5b files were written by torch 1.4 (docker/jukebox-embed.dockerfile:42): the pre-zipfile container, and
    -- ``fp16_params=True`` -- every prior ``Conv1D.w`` as a HalfTensor.  Both must come through the restricted reader
    unchanged (TopPrior rounds fp32 inputs to fp16 itself; fp16 inputs are taken as they are)."""
    hps = hparams_tiny()
    vq_ck, pr_ck, w = upstream_style_checkpoints(hps, ckpt_depth=hps.prior_depth)
    pr_ck["model"] = {k: (v.half() if k.endswith((".c_attn.w", ".c_proj.w", ".c_fc.w")) else v) for k, v in pr_ck["model"].items()}
    pv, pp = tmp_path / "vqvae.pth.tar", tmp_path / "prior_level_2.pth.tar"
    torch.save(vq_ck, pv, _use_new_zipfile_serialization=False)
    torch.save(pr_ck, pp, _use_new_zipfile_serialization=False)
    import zipfile

    assert not zipfile.is_zipfile(pv) and not zipfile.is_zipfile(pp)
    got, _ = CK.load_checkpoint_weights("5b", hps, None, str(pv), str(pp))
    n_half = 0
    for k, v in got.items():
        if k.endswith((".c_attn.w", ".c_proj.w", ".c_fc.w")):
            assert v.dtype == torch.float16 and torch.equal(v, w[k].half()), k
            n_half += 1
        else:
            assert v.dtype == w[k].dtype and torch.equal(v, w[k]), k
    assert n_half == 4 * hps.prior_depth
