"""GPU parity: VQ-VAE level-2 encoder + codebook vs the bit-exact C oracle (integer work -> exact)."""
import numpy as np
import pytest
import torch

from conftest import report_close
from llark_amd.jukebox.hparams import hparams_5b, hparams_tiny
from llark_amd.jukebox.synthetic import init_codebook_from_encodings, make_vqvae_weights, synthetic_clip

pytestmark = pytest.mark.gpu


def _clip(hps, i, seconds):
    from oracle import jukebox_ref as R
    a = R.normalize_audio(synthetic_clip(i, seconds=seconds))[: hps.sample_length]
    return np.pad(a, (0, max(0, hps.sample_length - len(a)))).astype(np.float32)


@pytest.fixture(scope="module")
def tiny_model():
    from llark_amd.jukebox.vqvae import VQVAE
    from oracle import jukebox_c as C
    hps = hparams_tiny()
    w = make_vqvae_weights(hps, 0)
    calib = np.concatenate([C.encoder_forward(w, _clip(hps, 100 + i, 1.6)[None], hps) for i in range(2)], axis=1)
    w["bottleneck.level_blocks.2.k"] = init_codebook_from_encodings(torch.from_numpy(calib), hps.l_bins)
    return hps, w, VQVAE(hps, w, "cuda")


def test_single_layers_bit_exact():
    """Every built conv shape, odd lengths, all dilations: bitwise equal to the C oracle."""
    from llark_amd import ops
    from oracle import jukebox_c as C
    g = torch.Generator().manual_seed(11)
    for (cin, cout, k, s, p, tin) in [(1, 32, 4, 2, 1, 1000), (32, 32, 4, 2, 1, 1030), (64, 32, 4, 2, 1, 514),
                                      (32, 64, 3, 1, 1, 129), (32, 64, 3, 1, 1, 2049)]:
        x = torch.randn(2, cin, tin, generator=g)
        wt = torch.randn(cout, cin, k, generator=g) * 0.2
        b = torch.randn(cout, generator=g)
        got = ops.conv1d(x.cuda(), ops.pack_conv_weight(wt.cuda()), b.cuda(), s, p).cpu().numpy()
        for n in range(2):
            ref = C.conv1d(x[n].numpy(), wt.numpy(), b.numpy(), stride=s, pad=p, dil=1)
            assert got[n].shape == ref.shape
            assert np.array_equal(got[n], ref), (
                f"conv cin={cin} cout={cout} k={k}: max diff {np.abs(got[n]-ref).max():.3e} "
                f"mismatches {(got[n]!=ref).sum()}/{ref.size}")
    for dil, t in [(1, 700), (3, 512), (9, 1025), (27, 600), (27, 40)]:
        x = torch.randn(2, 32, t, generator=g)
        w1 = torch.randn(32, 32, 3, generator=g) * 0.1
        b1 = torch.randn(32, generator=g) * 0.1
        w2 = torch.randn(32, 32, 1, generator=g) * 0.2
        b2 = torch.randn(32, generator=g) * 0.1
        got = ops.resblock(x.cuda(), ops.pack_conv_weight(w1.cuda()), b1.cuda(), ops.pack_conv_weight(w2.cuda()),
                           b2.cuda(), dil).cpu().numpy()
        for n in range(2):
            ref = C.resblock(x[n].numpy(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), dil)
            assert np.array_equal(got[n], ref), (
                f"resblock dil={dil} t={t}: max diff {np.abs(got[n]-ref).max():.3e} mismatches {(got[n]!=ref).sum()}")


def test_codebook_bit_exact_with_ties():
    from llark_amd import ops
    from oracle import jukebox_c as C
    g = torch.Generator().manual_seed(5)
    k = torch.randn(2048, 64, generator=g)
    k[77] = k[1500]                       # exact tie: the lower index must win
    x = torch.randn(2, 64, 300, generator=g)
    x[0, :, 10] = k[1500]                 # token 10 sits exactly on the duplicated code
    kd = k.cuda()
    codes, dist = ops.codebook_argmin(x.cuda(), kd, ops.codebook_norms(kd), want_dist=True)
    for n in range(2):
        ref_c, ref_d = C.codebook(x[n].numpy(), k.numpy())
        assert np.array_equal(codes[n].cpu().numpy(), ref_c)
        assert np.array_equal(dist[n].cpu().numpy(), ref_d)
    assert codes[0, 10].item() == 77


def test_encoder_and_codes_tiny(tiny_model):
    from oracle import jukebox_c as C
    hps, w, vq = tiny_model
    audio = np.stack([_clip(hps, i, 1.6) for i in range(3)])
    taps = []
    xe = vq.encoder_forward(torch.from_numpy(audio).cuda()[:, None, :], taps=taps)
    for n in range(3):
        ref_taps = []
        ref = C.encoder_forward(w, audio[n][None], hps, taps=ref_taps)
        for li, (a, b) in enumerate(zip(taps, ref_taps)):
            assert np.array_equal(a[n].cpu().numpy(), b), f"clip {n} layer {li}: max diff {np.abs(a[n].cpu().numpy()-b).max():.3e}"
        assert np.array_equal(xe[n].cpu().numpy(), ref)
    codes = vq.encode_top(torch.from_numpy(audio).cuda()).cpu().numpy()
    ref_codes = C.encode_codes(w, audio, hps)
    assert codes.shape == (3, hps.n_ctx)
    assert np.array_equal(codes, ref_codes), f"{(codes != ref_codes).sum()} code mismatches"
    zs = vq.encode(torch.from_numpy(audio[:1, :, None]).cuda())
    assert zs[0] is None and np.array_equal(zs[-1].cpu().numpy(), ref_codes[:1])


def test_ragged_and_error_inputs(tiny_model):
    hps, w, vq = tiny_model
    with pytest.raises(AssertionError):
        vq.encode_top(torch.zeros(1, hps.sample_length - 1, device="cuda"))
    from llark_amd import _lib
    with pytest.raises(_lib.LlarkHipError):
        vq.encode_top(torch.zeros(1, hps.sample_length))          # CPU tensor: no fallback
    # silence clip (all-zero): codes are still a valid constant sequence and equal the oracle's
    from oracle import jukebox_c as C
    z = vq.encode_top(torch.zeros(1, hps.sample_length, device="cuda")).cpu().numpy()
    assert np.array_equal(z, C.encode_codes(w, np.zeros((1, hps.sample_length), np.float32), hps))


def test_full_size_clip_exact():
    """BASELINE config size (1 048 576 samples -> 8192 codes): exact match with the C oracle."""
    from llark_amd.jukebox.vqvae import VQVAE
    from oracle import jukebox_c as C
    hps = hparams_5b()
    w = make_vqvae_weights(hps, 0)
    calib = C.encoder_forward(w, _clip(hps, 100, 25.0)[None], hps)
    w["bottleneck.level_blocks.2.k"] = init_codebook_from_encodings(torch.from_numpy(calib), hps.l_bins)
    vq = VQVAE(hps, w, "cuda")
    audio = _clip(hps, 0, 25.0)[None]
    codes, dist = vq.encode_top(torch.from_numpy(audio).cuda(), want_dist=True)
    ref, _, ref_d = C.encode_codes(w, audio, hps, return_all=True)
    assert np.array_equal(codes.cpu().numpy(), ref), f"{(codes.cpu().numpy() != ref).sum()} / {ref.size} code mismatches"
    assert np.array_equal(dist.cpu().numpy(), ref_d)
    assert len(np.unique(ref)) > 200


def test_fused_stage_weight_fragments():
    """llark_vqvae_pack_frag16: MFMA A fragments (hi / lo fp16) of a Conv1d weight, checked element by element against the layout
    the header states, including the accumulator-register channel order of the block's 1x1 convolution."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(3)
    for cout, cin, k, perm, e in ((32, 32, 3, False, 0), (32, 64, 4, False, 9), (64, 32, 3, False, -2), (32, 32, 1, True, 13)):
        w = torch.randn(cout, cin, k, generator=g) * 0.3
        hi, lo = ops.vqvae_pack_frag16(w.cuda(), perm1x1=perm, exp2=e)
        assert 2.0 ** 13 < float(w.abs().max()) * 2.0 ** ops.vqvae_weight_exponent(w) <= 2.0 ** 14
        w = w * 2.0 ** e                                               # the planes hold w 2^exp2 (power of two: exact)
        nks = k * cin // 16
        hi, lo = hi.cpu().view(cout // 32, nks, 64, 8), lo.cpu().view(cout // 32, nks, 64, 8)
        want = torch.empty(cout // 32, nks, 64, 8)
        for hb in range(cout // 32):
            for q in range(nks):
                tap, s = divmod(q, cin // 16)
                for lane in range(64):
                    co, gg = hb * 32 + (lane & 31), lane >> 5
                    for j in range(8):
                        c = 16 * s + (j & 3) + 8 * (j >> 2) + 4 * gg if perm else 16 * s + 8 * gg + j
                        want[hb, q, lane, j] = w[co, c, tap]
        assert torch.equal(hi.float(), want.half().float())
        assert torch.equal(lo.float(), (want - want.half().float()).half().float())


def test_fused_encoder_matches_exact_path(tiny_model):
    """The default encoder (csrc/vqvae_fused.hip: one launch per down-sampling step, split-fp16 MFMA products, activations in
    LDS / registers) against the per-layer fp32 path that is bit-equal to the C oracle: activations fp32-class (printed), VQ
    codes EQUAL -- for a batch, for a single clip (other window sizes) and for silence."""
    from oracle import jukebox_c as C
    hps, w, vq = tiny_model
    assert vq.exact is False and len(vq.stages) == sum(hps.downs_t)
    audio = np.stack([_clip(hps, i, 1.6) for i in range(3)])
    a = torch.from_numpy(audio).cuda()
    fused = vq.encoder_forward_fused(a)
    exact = vq.encoder_forward(a[:, None, :])
    assert fused.shape == exact.shape == (3, hps.emb_width, hps.n_ctx)
    scale = float(exact.abs().max())
    err = float((fused - exact).abs().max())
    print(f"\n[vqvae fused] tiny: max|fused - exact| {err:.3e} = {err / scale:.2e} of max|x| {scale:.3f}")
    assert err <= 2e-5 * scale, f"fused encoder output differs from the fp32 path by {err:.3e} ({err / scale:.2e} of max)"
    ref_codes = C.encode_codes(w, audio, hps)
    assert np.array_equal(vq.encode_top(a).cpu().numpy(), ref_codes)
    assert np.array_equal(vq.encode_top(a[1:2]).cpu().numpy(), ref_codes[1:2])
    one = vq.encoder_forward_fused(a[2:3])
    assert torch.equal(one[0], fused[2]), "a clip's encoding depends on the batch it rides in"


def test_fused_encoder_full_size_clip_codes():
    """BASELINE config size through the fused path: 8192 codes equal the C oracle's; the activation difference to the exact fp32
    path is orders of magnitude below the smallest best / second-best codebook gap of the clip."""
    from llark_amd.jukebox.vqvae import VQVAE
    from oracle import jukebox_c as C
    hps = hparams_5b()
    w = make_vqvae_weights(hps, 0)
    calib = C.encoder_forward(w, _clip(hps, 100, 25.0)[None], hps)
    w["bottleneck.level_blocks.2.k"] = init_codebook_from_encodings(torch.from_numpy(calib), hps.l_bins)
    vq = VQVAE(hps, w, "cuda")
    audio = np.stack([_clip(hps, 0, 25.0), _clip(hps, 7, 25.0)])
    a = torch.from_numpy(audio).cuda()
    codes = vq.encode_top(a).cpu().numpy()
    ref, _, ref_d = C.encode_codes(w, audio[:1], hps, return_all=True)
    assert np.array_equal(codes[:1], ref), f"{(codes[:1] != ref).sum()} / {ref.size} code mismatches"
    fused = vq.encoder_forward_fused(a[:1])
    exact = vq.encoder_forward(a[:1, None, :])
    scale, err = float(exact.abs().max()), float((fused - exact).abs().max())
    _, dist = vq.encode_top(a[:1], want_dist=True)
    print(f"\n[vqvae fused] full-size clip: max|fused - exact| {err:.3e} = {err / scale:.2e} of max|x| {scale:.3f}")
    assert err <= 2e-5 * scale
    exact_codes = VQVAE(hps, w, "cuda", exact=True).encode_top(a)
    assert np.array_equal(exact_codes.cpu().numpy(), codes)


# ---------------------------------------------------------------------------------------------
# round 4: near-tie certificate + exact window fix-up
# ---------------------------------------------------------------------------------------------
def test_near_tie_fixup_reproduces_exact_codes_when_everything_is_flagged(tiny_model):
    """With a threshold that flags EVERY token, all codes come from the exact kernels run on receptive-field windows
    (4 passes of 256 windows; windows on both clip edges and in the interior): they must equal the exact path's / the C
    oracle's codes -- the property the fix-up rests on (position-independent fmaf chains => window value == clip value)."""
    from llark_amd.jukebox.vqvae import VQVAE
    from oracle import jukebox_c as C
    hps, w, _ = tiny_model
    audio = np.stack([_clip(hps, i, 1.6) for i in range(2)])
    audio[1, 3000:] = 0.0                                  # silence to the end of the clip: exact ties in the tail
    a = torch.from_numpy(audio).cuda()
    vq = VQVAE(hps, w, "cuda", tie_e_rel=1e3)
    assert vq.win_tokens < hps.n_ctx and vq.win_tokens >= 2 * vq.halo_tokens + 1
    codes = vq.encode_top(a)
    assert vq.last_near_ties == 2 * hps.n_ctx
    ref = C.encode_codes(w, audio, hps)
    assert np.array_equal(codes.cpu().numpy(), ref), f"{(codes.cpu().numpy() != ref).sum()} window codes differ from the C oracle"
    # certificate off: the plain fused argmin (round 3's behaviour) still runs
    vq0 = VQVAE(hps, w, "cuda", tie_e_rel=None)
    c0 = vq0.encode_top(a)
    assert c0.shape == codes.shape and vq0.last_near_ties == 0
    # default thresholds: few tokens flagged, codes equal
    vqd = VQVAE(hps, w, "cuda")
    cd = vqd.encode_top(a)
    assert np.array_equal(cd.cpu().numpy(), ref)
    print(f"\n[near-tie] tiny: default thresholds flag {vqd.last_near_ties} of {2 * hps.n_ctx} tokens")


def test_near_tie_list_overflow_goes_exact():
    """More flagged tokens than the device list holds (8192 > 4096 at 5b size): the batch is re-encoded by the exact path."""
    from llark_amd.jukebox import vqvae as V
    from oracle import jukebox_c as C
    hps = hparams_5b()
    w = make_vqvae_weights(hps, 0)
    calib = C.encoder_forward(w, _clip(hps, 100, 25.0)[None], hps)
    w["bottleneck.level_blocks.2.k"] = init_codebook_from_encodings(torch.from_numpy(calib), hps.l_bins)
    a = torch.from_numpy(_clip(hps, 1, 25.0)[None]).cuda()
    vq = V.VQVAE(hps, w, "cuda", tie_e_rel=1e3)
    codes = vq.encode_top(a)
    assert vq.last_near_ties == hps.n_ctx > V.TIE_LIST_CAP
    assert torch.equal(codes, V.VQVAE(hps, w, "cuda", exact=True).encode_top(a))


def test_default_encoder_codes_equal_c_oracle_on_bench_clips():
    """THE parity bar of the benchmarked encoder (integer work -> exact): the default path (fused stages + near-tie
    certificate) on bench.py's own 8 clips (rank 0: clips 0..7, one batch of 8 like the bench) + 2 more gives the C oracle's
    codes (tests/golden/jukebox_full36_wide.npz, made by tests/golden/make_jukebox_wide_golden.py) with 0 mismatches.  The
    fused argmin alone (certificate off) is counted next to it."""
    import fulldepth as FD
    from llark_amd.jukebox.vqvae import VQVAE
    z = np.load(FD.WIDE_NPZ)
    hps = FD.jukebox_hps()
    w = make_vqvae_weights(hps, 0)
    vq = VQVAE(hps, w, "cuda")
    cal = torch.from_numpy(FD.jukebox_clip(FD.CAL_CLIP, hps)).cuda()[None, None, :]
    k = FD.codebook_from_encoding(vq.encoder_forward(cal)[0].cpu(), hps)
    assert FD.sha(k.numpy()) == str(z["codebook_sha"])
    vq.set_codebook(k)
    raw = VQVAE(hps, {**w, "bottleneck.level_blocks.2.k": k}, "cuda", tie_e_rel=None)
    clips = [int(c) for c in z["codes10_clips"]]
    gold = z["codes10"].astype(np.int64)
    total_flag = total_raw_mismatch = 0
    for lo, hi in ((0, 8), (8, 10)):
        a = torch.from_numpy(np.stack([FD.jukebox_clip(i, hps) for i in clips[lo:hi]])).cuda()
        got = vq.encode_top(a).cpu().numpy()
        bad = int((got != gold[lo:hi]).sum())
        r = int((raw.encode_top(a).cpu().numpy() != gold[lo:hi]).sum())
        print(f"\n[near-tie] clips {clips[lo:hi]}: {vq.last_near_ties} of {got.size} tokens re-evaluated exactly; mismatches vs the C oracle: "
              f"{bad} (fused argmin without the certificate: {r})")
        total_flag += vq.last_near_ties
        total_raw_mismatch += r
        assert bad == 0, f"{bad} codes differ from the C oracle on clips {clips[lo:hi]}"
    assert total_flag < 0.01 * gold.size, "the certificate flags more than 1 % of the tokens: thresholds far too wide"
