"""GPU parity AT THE BENCHMARKED CONFIGURATION (VERDICT r01 "next round" item 1).

bench.py times 36 prior layers x B = 8 clips and 32 Llama-2-7B layers at S = 371; these tests compare exactly that
depth / width / batch with the CPU oracle's committed outputs:

  tests/golden/jukebox_full36.npz   <- tests/golden/make_jukebox_full_golden.py   (oracle/jukebox_ref.py + jukebox_ref.c)
  tests/golden/llama7b_full32.npz   <- tests/golden/make_llama7b_golden.py        (oracle/llama_ref.py, fp32)

Bars (BASELINE.json): VQ codes exact; embeddings max-abs-err <= 1e-4 ABSOLUTE for the default precision f16x2 (configs[1],
SURVEY 8(d); the opt-in lo8 mode only meets 1e-4 * max|acts| and says so); logits max|err| <= 1e-3 * max|logits|
(north_star); 64 greedy tokens equal (configs[2]).  Weights are regenerated from the CPU seeds of
tests/fulldepth.py on both sides.  The measured errors are printed (run with -s to see them).
"""
import numpy as np
import pytest
import torch

import fulldepth as FD
from conftest import report_close
from llark_amd.jukebox.prior import DEFAULT_PRECISION

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["f16x2", "lo8"])
def jb(request):
    """Both precisions of the prior's Conv1D products (llark_amd/jukebox/prior.py): "f16x2" = fp16 hi + fp16 lo planes
    (22 significant bits), "lo8" = fp16 hi + E4M3 lo plane (15-16 bits, one MX-fp8 MFMA instead of the second fp16 pass)."""
    from llark_amd.jukebox import extract as E

    z = np.load(FD.JUKEBOX_NPZ)
    hps = FD.jukebox_hps()
    w = FD.jukebox_weights_cpu(hps)
    enc = E.WrappedAudioEncoder(hps=hps, weights=w, device="cuda", precision=request.param)
    assert enc.top_prior.prior.precision == request.param
    # data-dependent codebook from the calibration clip through the HIP encoder; its checksum equals the one the C
    # oracle produced in the build container <=> the full-size encoder output is bit-exact
    cal = torch.from_numpy(FD.jukebox_clip(FD.CAL_CLIP, hps)).cuda()[None, None, :]
    k = FD.codebook_from_encoding(enc.vqvae.encoder_forward(cal)[0].cpu(), hps)
    assert FD.sha(k.numpy()) == str(z["codebook_sha"]), "calibration-clip encoder output differs from the C oracle's"
    enc.vqvae.set_codebook(k)
    del w
    yield z, hps, enc
    del enc
    torch.cuda.empty_cache()


def test_jukebox_36_layers_batch8_vs_oracle(jb):
    """configs[1]: batch = 8 x 25 s clips, all 36 layers; the pinned clip sits at batch row 5."""
    z, hps, enc = jb
    order = [1, 2, 3, 4, 5, FD.GOLD_CLIP, 6, 7]
    row = order.index(FD.GOLD_CLIP)
    a0 = FD.jukebox_clip(FD.GOLD_CLIP, hps)
    assert FD.sha(a0) == str(z["audio_sha"])
    audio = torch.from_numpy(np.stack([FD.jukebox_clip(i, hps) for i in order])).cuda()
    codes = enc.vqvae.encode_top(audio)
    assert np.array_equal(codes[row].cpu().numpy(), z["codes"].astype(np.int64)), "VQ codes differ from the oracle (36-layer fixture)"
    emb = enc(audio)
    assert emb.shape == (8, 240, hps.prior_width)
    prec = enc.top_prior.prior.precision
    ref = torch.from_numpy(z["emb_f10"]).double()
    got = emb[row].cpu().double()
    err = float((got - ref).abs().max())
    acts_max, emb_max, emb_rms = float(z["acts_maxabs"]), float(ref.abs().max()), float(ref.pow(2).mean().sqrt())
    # The error three ways: absolute, / max|emb|, / rms|emb| (and / max|un-pooled acts|, round 2's normaliser)
    print(f"\n[fulldepth] jukebox 36 layers x B=8 ({prec}): embedding max|err| ABS {err:.3e} | / max|emb| {emb_max:.2f} = {err / emb_max:.2e} "
          f"| / rms|emb| {emb_rms:.2f} = {err / emb_rms:.2e} | / max|acts| {acts_max:.2f} = {err / acts_max:.2e}")
    if prec == "f16x2":
        # the library default: BASELINE configs[1] / SURVEY 8(d) read literally -- max-abs-err <= 1e-4, no normaliser
        assert DEFAULT_PRECISION == "f16x2"
        assert err <= 1e-4, f"default precision: embedding max-abs-err {err:.3e} > 1e-4"
    else:
        # opt-in reduced-precision mode: meets the bar only relative to max|acts| (15-16 bit activations; 5.1e-4 absolute in round 2)
        assert err <= 1e-4 * acts_max, f"lo8: embedding max|err| {err:.3e} > 1e-4 * max|acts| = {1e-4 * acts_max:.3e}"
        assert err <= 1e-3, "lo8: error far above its measured level"
    # batch invariance: the same clip alone (B = 1) gives bit-identical codes and embedding
    one = torch.from_numpy(a0).cuda()[None]
    assert torch.equal(enc.vqvae.encode_top(one)[0], codes[row])
    emb1 = enc(one)
    assert torch.equal(emb1[0], emb[row]), "B=1 and B=8 embeddings of the same clip are not bit-identical"
    # every row of the batch is finite and distinct
    assert torch.isfinite(emb).all() and not torch.equal(emb[0], emb[1])


def test_jukebox_36_layers_error_growth_and_global_mean(jb):
    """Un-pooled residual-stream rows at depth 1, 3, 6, 12, 24, 36 (the measured error of the hi+lo fp16 scheme as it
    compounds), and the f = 0 global-mean branch (jukebox/main.py:158-159)."""
    from llark_amd import ops
    from llark_amd.jukebox import extract as E

    z, hps, enc = jb
    tp = enc.top_prior
    codes = torch.from_numpy(z["codes"].astype(np.int64))[None].cuda()
    x_cond, y_cond = E.get_cond(hps, tp)
    tp.prior.only_encode = True
    h = tp.prior.embed(codes, x_cond[0:1], y_cond)
    h2 = h.view(hps.n_ctx, hps.prior_width)
    rows = torch.from_numpy(z["probe_rows"]).cuda()
    layers = [int(v) for v in z["probe_layers"]]
    report = []
    for d in range(hps.prior_depth):
        tp.prior.layer_forward(h2, d, 1)
        if d + 1 in layers:
            i = layers.index(d + 1)
            scale = float(z["maxabs"][i])
            err = report_close(f"probe rows after layer {d + 1}", h2[rows].cpu(), z["probes"][i], 1e-4 * scale)
            report.append((d + 1, err, err / scale))
    print(f"\n[fulldepth] prior error growth, {tp.prior.precision} (un-pooled probe rows: max|err| absolute, and / max|h|): "
          + ", ".join(f"L{l}: {a:.2e} ({e:.2e})" for l, a, e in report))
    mean = ops.pool_mean(h2.contiguous()[None])[0]
    strict = tp.prior.precision == "f16x2"        # default precision: absolute 1e-4 on the pooled outputs
    report_close("36-layer global mean (f=0)", mean.cpu(), z["emb_f0"], 1e-4 if strict else 1e-4 * float(z["acts_maxabs"]))


def test_jukebox_near_tie_audit_fixture():
    """Parity honesty (VERDICT weak #2): on the full-size clip the defined-order C oracle, the order-free torch
    F.conv1d restatement and a float64 argmin agree on every code, and the smallest best/second-best codebook gap
    is far above the fp32 accumulation-order noise of the encoder output."""
    z = np.load(FD.JUKEBOX_NPZ)
    assert float(z["agree_torch"]) == 1.0 and float(z["agree_f64"]) == 1.0
    assert np.array_equal(z["codes"], z["codes_torch"])
    # an order change moves an encoder output by <= enc_maxdiff; the distance moves by <= 2*|x-k|*sqrt(64)*diff
    assert float(z["gap"].min()) > 50 * float(z["enc_maxdiff_torch_vs_c"])


# ---------------------------------------------------------------------------------------------
# round 4: the wide fixture -- a second spectrum, a 12 s clip, outlier weights, float64 noise floor
# ---------------------------------------------------------------------------------------------
def _wide_case_report(name, z, got_f10, got_f0, codes):
    """Errors of one case against the fp32 oracle (the reference-class CPU path) and against the float64 evaluation of the same
    graph over the first frames (the truth both approximate), printed; returns (abs err vs fp32 oracle, err vs float64,
    the oracle's own err vs float64)."""
    # round 6 (VERDICT r05 hygiene): the fixture keeps every 4th pooled frame + the last one (`_rows`) of each case and five of the first
    # 30 float64 frames -- each frame is a mean over 34 tokens of all 4800 channels of the 36-layer output, so a wrong kernel shows in all
    ref = z[f"{name}_emb_f10"].astype(np.float64)
    rows, frames = z[f"{name}_emb_f10_rows"], int(z[f"{name}_emb_f10_frames"])
    assert got_f10.shape == (frames, ref.shape[1]), f"{name}: embedding shape {got_f10.shape}, oracle ({frames}, {ref.shape[1]})"
    assert np.array_equal(codes, z[f"{name}_codes"].astype(np.int64)), f"{name}: VQ codes differ from the C oracle"
    err = float(np.abs(got_f10[rows] - ref).max())
    err0 = float(np.abs(got_f0 - z[f"{name}_emb_f0"].astype(np.float64)).max())
    h64, h64_frames = z[f"head64_{name}_f10"], z[f"head64_{name}_f10_frames"]
    nf = int(h64_frames[-1]) + 1
    err64 = float(np.abs(got_f10[h64_frames] - h64).max())
    orc64 = float(z[f"head64_{name}_fp32_oracle_err"])
    emb_max = float(z[f"{name}_emb_f10_maxabs"])
    print(f"\n[fulldepth wide] {name}: frames {frames}, max|emb| {emb_max:.2f}, max|acts| {float(z[f'{name}_acts_maxabs']):.2f} | HIP vs fp32 oracle: "
          f"f=10 max|err| {err:.3e} ({err / emb_max:.2e} of max|emb|), f=0 {err0:.3e} | first {nf} frames vs float64: HIP {err64:.3e}, "
          f"fp32 oracle itself {orc64:.3e}")
    return err, err64, orc64


def test_jukebox_36_layers_more_clips_vs_oracle(jb):
    """A clip with a different spectrum and a 12 s clip (latent_audio_len = 4134 -> 121 frames: the slice of
    jukebox/main.py:154 and a frame count != 240), one batch of two through the reference-shaped batch entry point:
    embedding max-abs-err <= 1e-4 in the default precision, codes exact; errors printed per clip, also against float64."""
    from llark_amd.jukebox import extract as E

    z0, hps, enc = jb
    if enc.top_prior.prior.precision != "f16x2":
        pytest.skip("the wide cases are run in the default precision")
    z = np.load(FD.WIDE_NPZ)
    assert str(z["codebook_sha"]) == str(z0["codebook_sha"])
    names = ["rich", "short12"]
    audios = [FD.jukebox_case_audio(*FD.WIDE_CASES[n][:3]) for n in names]
    assert len(audios[1]) < hps.sample_length < len(audios[0])
    f10 = E.get_acts_from_audio_batch(audios, hps, enc.vqvae, enc.top_prior, meanpool=True, pool_frames_per_second=10)
    f0 = E.get_acts_from_audio_batch(audios, hps, enc.vqvae, enc.top_prior, meanpool=True, pool_frames_per_second=None)
    for i, name in enumerate(names):
        a = np.pad(audios[i], (0, max(0, hps.sample_length - len(audios[i]))))[: hps.sample_length].astype(np.float32)
        assert FD.sha(a) == str(z[f"{name}_audio_sha"])
        codes = enc.vqvae.encode_top(torch.from_numpy(a).cuda()[None])[0].cpu().numpy()
        err, err64, orc64 = _wide_case_report(name, z, f10[i].astype(np.float64), f0[i].astype(np.float64), codes)
        assert f10[i].shape[0] == int(z[f"{name}_latent_len"]) // enc.frame_len
        assert err <= 1e-4, f"{name}: embedding max-abs-err {err:.3e} > 1e-4"
    assert f10[1].shape[0] == 121 and f10[0].shape[0] == 240
    # the pinned clip of round 2 against float64 too: how much of its 5e-5 is the fp32 oracle's own rounding
    base = enc(torch.from_numpy(FD.jukebox_clip(FD.GOLD_CLIP, hps)).cuda()[None])[0].cpu().double().numpy()
    h64, h64_frames = z["head64_base_f10"], z["head64_base_f10_frames"]
    e64 = float(np.abs(base[h64_frames] - h64).max())
    print(f"\n[fulldepth wide] base clip 0, frames {h64_frames.tolist()} vs float64: HIP {e64:.3e}, fp32 oracle itself {float(z['head64_base_fp32_oracle_err']):.3e}")
    assert e64 <= 1e-4


def test_jukebox_36_layers_outlier_weights_vs_oracle():
    """Robustness: the same 4 hidden channels are x30 outliers in every layer's residual-writing
    columns (tests/fulldepth.add_outlier_channels): |h| reaches ~280 next to O(1) channels, max|emb| 165.  The fp32 CPU
    oracle itself is 7e-4 away from the float64 evaluation of the graph there, so the absolute 1e-4 of configs[1] cannot be
    the bar; asserted: the HIP path is no further from float64 than 2x the fp32 oracle is, and within 1e-5 of max|emb| of
    the fp32 oracle.  The absolute numbers are printed: they are the finding."""
    from llark_amd.jukebox import extract as E

    z = np.load(FD.WIDE_NPZ)
    hps = FD.jukebox_hps()
    w = FD.jukebox_weights_cpu(hps)
    ch = FD.add_outlier_channels(w, hps)
    assert ch == [int(c) for c in z["outlier_channels"]]
    enc = E.WrappedAudioEncoder(hps=hps, weights=w, device="cuda", precision="f16x2")
    cal = torch.from_numpy(FD.jukebox_clip(FD.CAL_CLIP, hps)).cuda()[None, None, :]
    k = FD.codebook_from_encoding(enc.vqvae.encoder_forward(cal)[0].cpu(), hps)
    assert FD.sha(k.numpy()) == str(z["codebook_sha"])
    enc.vqvae.set_codebook(k)
    del w
    a = FD.jukebox_case_audio(*FD.WIDE_CASES["outlier"][:3])
    f10 = E.get_acts_from_audio_batch([a], hps, enc.vqvae, enc.top_prior, meanpool=True, pool_frames_per_second=10)[0]
    f0 = E.get_acts_from_audio_batch([a], hps, enc.vqvae, enc.top_prior, meanpool=True, pool_frames_per_second=None)[0]
    ap = np.pad(a, (0, max(0, hps.sample_length - len(a))))[: hps.sample_length].astype(np.float32)
    codes = enc.vqvae.encode_top(torch.from_numpy(ap).cuda()[None])[0].cpu().numpy()
    err, err64, orc64 = _wide_case_report("outlier", z, f10.astype(np.float64), f0.astype(np.float64), codes)
    emb_max = float(z["outlier_emb_f10_maxabs"])
    assert np.isfinite(f10).all()
    assert err64 <= 2.0 * orc64, f"outlier weights: HIP is {err64:.3e} from float64, the fp32 oracle {orc64:.3e}"
    assert err <= 1e-5 * emb_max, f"outlier weights: {err:.3e} from the fp32 oracle = {err / emb_max:.2e} of max|emb|"
    del enc
    torch.cuda.empty_cache()


# Bars of the two activation flows of the Llama half at FULL depth.  "split" = the default / headline flow
# (fp32-class, bar = north_star's 1e-3 of max|logits|).  "bf16" = single bf16 rounding at every Linear input / q / k / v / attention
# output (the dtype FLOW of the reference's GPU path, which itself runs 16-bit: m2t/models/utils.py:129 torch_dtype=float16): its
# error against the fp32 oracle is MEASURED and recorded here and in profiles/r05_llama_fulldepth_parity.json; the assertion is a
# regression guard at 2x the recorded figure, not a claim that the flow meets 1e-3.
LLM_BARS = {"split": 1e-3, "bf16": 8e-2}     # bf16: measured 3.7e-2 of max|logits| (round 5, profiles/r05_llama_fulldepth_parity.json)
_LLM_REPORT = {}


def _llm_report(precision, **kv):
    """Collects the measured figures per precision; written to gpurun_out/ (copied to profiles/ by the run script)."""
    import json
    import os

    _LLM_REPORT.setdefault(precision, {}).update(kv)
    out = os.path.join(FD.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "llama_fulldepth_parity.json"), "w") as f:
        json.dump(_LLM_REPORT, f, indent=1, sort_keys=True)


@pytest.fixture(scope="module", params=["split", "bf16"])
def llm(request):
    from llark_amd.m2t.engine import HipLlamaEngine, LlamaDims

    z = np.load(FD.LLAMA_NPZ)
    spec = FD.llama_spec(32)
    w = FD.llama_weights_cpu(spec)
    dims = LlamaDims(vocab_size=spec.vocab_size)
    eng = HipLlamaEngine(dims, "cuda", max_batch=8, max_seq=512, precision=request.param)
    eng.load_state_dict(w)
    del w
    yield z, spec, eng
    del eng
    torch.cuda.empty_cache()


def test_llama7b_32_layers_logits_vs_oracle(llm):
    """32 layers, S = 371, B = 8 (bench batch; every row the pinned prompt + audio): logits <= 1e-3 * max|logits|."""
    z, spec, eng = llm
    ids, aud = FD.llama_inputs(1)
    assert FD.sha(ids.numpy()) == str(z["ids_sha"]) and FD.sha(aud.numpy()) == str(z["aud_sha"])
    B = 8
    ids8 = ids.expand(B, -1).contiguous().cuda()
    audc = aud[0].cuda()
    logits = eng.forward_tokens(ids8, [(b, 1, audc) for b in range(B)])
    rows = torch.from_numpy(z["rows"]).cuda()
    scale = float(z["logits_maxabs"])
    prec = eng.precision
    got, ref = logits[3][rows].cpu().double(), torch.from_numpy(z["logits_rows"]).double()
    err = float((got - ref).abs().max())
    rms = float((got - ref).pow(2).mean().sqrt())
    agree = float((got.argmax(-1) == ref.argmax(-1)).double().mean())
    print(f"\n[fulldepth] llama-2-7B 32 layers S=371 B=8 ({prec}): logits max|err| {err:.3e} = {err / scale:.2e} of max|logits| {scale:.3f}; "
          f"rms err {rms:.3e}; argmax of the sampled rows equal to the oracle's: {agree:.4f}")
    _llm_report(prec, logits_max_abs_err=err, logits_max_abs=scale, logits_err_over_max=err / scale, logits_rms_err=rms,
                sampled_rows_argmax_agree=agree, bar_over_max=LLM_BARS[prec], layers=32, seq=371, batch=8)
    report_close(f"7B 32-layer logits (sampled rows, full vocab) vs fp32 oracle [{prec}]", got, ref, LLM_BARS[prec] * scale)
    # rows of the batch hold the same prompt: bit-identical results (batch invariance of every kernel)
    assert torch.equal(logits[0], logits[7])
    # B = 1 goes through different tile counts and K cuts (o_proj / down_proj: 4 K ranges per tile instead of 2); same values
    # to the rounding of the fp32 accumulation and of the bf16 hi/lo operand planes downstream (measured 3.1e-5 of max|logits|)
    l1 = eng.forward_tokens(ids.cuda(), [(0, 1, audc)])
    # (bf16 flow: a K cut that moves changes which fp32 sum gets rounded to bf16 downstream -- differences of the size of the flow's own error)
    report_close("B=1 vs B=8 logits", l1[0][rows].cpu(), logits[0][rows].cpu(), (6e-5 if prec == "split" else LLM_BARS[prec]) * scale)


def test_llama7b_64_greedy_tokens_vs_oracle(llm):
    """configs[2]: prefill + 64 greedy decode steps against the KV cache.  "split": tokens equal the oracle's, and the
    last-position logits of decode steps 0 / 1 / 31 / 63 stay within 1e-3 * max|logits|.  "bf16": the number of the 64 positions
    whose argmax equals the oracle's token is COUNTED (teacher-forced on the oracle's tokens, so that one early flip does not hide
    the other 63 comparisons) and recorded with the worst checked decode-logit error; positions that differ must be near-ties of
    the oracle (its top-1 / top-2 gap below twice the flow's measured logit error)."""
    z, spec, eng = llm
    ids, aud = FD.llama_inputs(1)
    audc = aud[0].cuda()
    gold, gaps = z["tokens"], z["gaps"]
    step_idx = [int(v) for v in z["step_idx"]]
    scale = float(z["logits_maxabs"])
    prec = eng.precision
    bar = LLM_BARS[prec] * scale
    eng.reset(1)
    logits = eng.forward_tokens(ids.cuda(), [(0, 1, audc)], last_only=True)
    toks, worst, flips = [], 0.0, []
    for t in range(len(gold)):
        last = logits[0, -1]
        if t in step_idx:
            e = float((last.cpu().double() - torch.from_numpy(z["step_logits"][step_idx.index(t)]).double()).abs().max())
            worst = max(worst, e)
        tok = int(last.argmax())
        toks.append(tok)
        if tok != int(gold[t]):
            if prec == "split":
                raise AssertionError(f"greedy token {t}: got {tok}, oracle {int(gold[t])} (oracle top-1/top-2 gap {gaps[t]:.3e}, "
                                     f"logit error bound so far {worst:.3e})")
            flips.append((t, float(gaps[t])))
        if t + 1 < len(gold):
            nxt = torch.tensor([[int(gold[t])]], device="cuda")          # == tok for "split"; teacher-forced for "bf16"
            logits = eng.forward_tokens(nxt, (), pos0=eng.cur_len, last_only=True)
    match = len(gold) - len(flips)
    _llm_report(prec, greedy_tokens_matching=match, greedy_tokens=len(gold), decode_logits_worst_err_over_max=worst / scale,
                greedy_flips=[{"step": t, "oracle_gap": g} for t, g in flips], oracle_min_gap=float(gaps.min()))
    print(f"\n[fulldepth] ({prec}) {match} of {len(gold)} greedy tokens equal the oracle's; smallest oracle top-1/top-2 gap {gaps.min():.3e}, "
          f"worst checked decode-logit error {worst:.3e} ({worst / scale:.2e} of max|logits|); flips (step, oracle gap): {flips}")
    assert worst <= bar, f"({prec}) worst checked decode-logit error {worst:.3e} > {bar:.3e}"
    if prec == "split":
        assert toks == [int(v) for v in gold]
    else:
        for t, g in flips:
            assert g <= 2.0 * bar, f"bf16 flow flips greedy token {t} although the oracle's gap {g:.3e} is far above the flow's error"
