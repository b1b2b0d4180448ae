"""CPU checks of the folded-LayerNorm algebra the HIP path uses (llark_gemm16_ln, llark_amd/jukebox/prior.py): no GPU, no library
compute call.  The GPU parity tests (tests/test_prior_gpu.py) compare the kernels with this same restatement in float64."""
import numpy as np
import torch

from oracle import jukebox_ref as R


def test_folded_layernorm_identity_against_oracle_ops():
    """LN(x) W + b  ==  rstd (x . gamma) W - rstd mean (gamma W) + (beta W + b), with the oracle's own layer_norm / Conv1D
    (oracle/jukebox_ref.py -- the functions the 36-layer fixtures were generated with) on the left-hand side."""
    g = torch.Generator().manual_seed(0)
    m, k, n = 37, 96, 80
    x = torch.randn(m, k, generator=g, dtype=torch.float64) * 3 + 0.5
    gamma = 1 + 0.3 * torch.randn(k, generator=g, dtype=torch.float64)
    beta = 0.2 * torch.randn(k, generator=g, dtype=torch.float64)
    w = torch.randn(k, n, generator=g, dtype=torch.float64) * 0.1            # upstream Conv1D.w: [n_in][n_out]
    b = torch.randn(n, generator=g, dtype=torch.float64)
    ln = torch.nn.functional.layer_norm(x, (k,), gamma, beta, 1e-5)
    want = R._conv1d_linear(ln, w, b, dtype=torch.float64)
    mean = x.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    gw = gamma @ w                                                            # ln_vec of the consumer
    bw = beta @ w + b                                                         # its bias
    got = rstd * ((x * gamma) @ w - mean * gw) + bw
    assert float((got - want).abs().max()) < 1e-12


def test_partial_sum_partition_matches_the_kernel_layout():
    """The producer writes, per row, one (sum, sum of squares) pair per 256-column tile and wave column wn: wave wn of tile j owns
    columns 256 j + 64 wn + {0 .. 63} and 256 j + 128 + 64 wn + {0 .. 63} (csrc/gemm256x.hip: two halves of 128 columns, the wave's
    64 columns in each).  Every column of [0, N) is in exactly one slice, also in the ragged last tile, so the slice-ordered sum
    llark_ln_stats_finalize forms is the row sum -- for ANY N % 4 == 0."""
    for n in (4800, 1216, 3648, 260, 4):
        tiles = (n + 255) // 256
        owner = -np.ones(n, dtype=np.int64)
        for j in range(tiles):
            for wn in range(2):
                for half in range(2):
                    c0 = 256 * j + 128 * half + 64 * wn
                    cols = np.arange(c0, min(c0 + 64, n))
                    assert (owner[cols] == -1).all()
                    owner[cols] = 2 * j + wn
        assert (owner >= 0).all() and owner.max() <= 2 * tiles - 1
        x = np.random.default_rng(n).standard_normal(n)
        parts = np.array([[x[owner == s].sum(), (x[owner == s] ** 2).sum()] for s in range(2 * tiles)])
        mean = parts[:, 0].sum() / n
        var = parts[:, 1].sum() / n - mean * mean
        assert abs(mean - x.mean()) < 1e-12 and abs(var - x.var()) < 1e-12


def test_fold_takes_only_big_tiles_signature_exported():
    """The C-ABI names the Python layer binds (no compute call): present in the header and in the ctypes table."""
    import os
    import re

    from llark_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "llark_hip.h")).read()
    for name in ("llark_gemm16_ln", "llark_gemm16_ln_takes", "llark_ln_stats_finalize"):
        assert re.search(r"\b%s\s*\(" % name, hdr), name
        assert name in _lib._SIGS, name


def _split16(v32):
    """fp16 hi / lo planes of an fp32 array (numpy float16 rounds to nearest even and keeps subnormals, like the kernels)."""
    hi = v32.astype(np.float16)
    lo = (v32 - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64) + lo.astype(np.float64)


def _pow2_near(v):
    m, e = np.frexp(v)
    e = np.where(m > 0.70710678, e, e - 1)
    return np.ldexp(1.0, np.clip(e, -24, 24))


def test_predicted_statistics_keep_the_planes_bits_on_adversarial_rows():
    """Round 5: what the folded planes lose on rows whose level / spread is far from one, and what pre-normalising them
    with PREDICTED statistics recovers (llark_gemm16_ln_p + llark_ln_stats_finalize_p, restated here in numpy with fp16 planes).
    Rows: std 1e-3 (the lo plane in its subnormals), std 1e3 with a mean of 50 sigma (hi plane overflows without the shift), mean of
    50 sigma at std 1.  The prediction is deliberately off (mean by 0.3 sigma, scale by a factor 1.7) -- it only has to be close.
    Identity checked on the way: with stat = ((mean - shift) scale, rstd / scale) the consumer formula is unchanged."""
    rng = np.random.default_rng(0)
    K, N = 4800, 128
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16).astype(np.float64)
    gamma = (1 + 0.3 * rng.standard_normal(K)).astype(np.float32)
    beta = 0.2 * rng.standard_normal(K)
    gw, bw = W @ gamma.astype(np.float64), W @ beta
    worst = {}
    for name, scale, off in (("unit", 1.0, 0.0), ("std 1e-3", 1e-3, 0.0), ("std 3e-3", 3e-3, 0.0), ("mean 50 sigma", 1.0, 50.0),
                             ("std 1e3, mean 50 sigma", 1e3, 50.0), ("std 1e-3, mean 50 sigma", 1e-3, 50.0)):
        x = (rng.standard_normal((16, K)) * scale + off * scale).astype(np.float32)
        x64 = x.astype(np.float64)
        mu, var = x64.mean(1, keepdims=True), x64.var(1, keepdims=True)
        rstd = 1.0 / np.sqrt(var + 1e-5)
        ref = ((x64 - mu) * rstd * gamma + beta) @ W.T
        with np.errstate(over="ignore", invalid="ignore"):
            plain = rstd * (_split16(x * gamma) @ W.T - mu * gw) + bw                      # round 4: planes of x . gamma
        shift = (mu + 0.3 / rstd).astype(np.float32)                                        # a prediction that is off
        sc = _pow2_near(1.7 * rstd).astype(np.float32)
        planes = _split16(((x - shift) * sc) * gamma)
        assert np.isfinite(planes).all(), name
        d = (x - shift).astype(np.float64)                                                  # what the producer sums
        dm = d.mean(1, keepdims=True)
        var_p = (d * d).mean(1, keepdims=True) - dm * dm
        rstd_p = 1.0 / np.sqrt(np.maximum(var_p, 0) + 1e-5)
        stat = (dm * sc, rstd_p / sc)                                                       # ((mean - shift) scale, rstd / scale)
        pred = stat[1] * (planes @ W.T - stat[0] * gw) + bw                                 # the UNCHANGED consumer formula
        s = np.abs(ref).max()
        worst[name] = (np.nanmax(np.abs(plain - ref)) / s if np.isfinite(plain).all() else np.inf, np.abs(pred - ref).max() / s)
        assert worst[name][1] <= 3e-7, f"{name}: predicted-statistics planes {worst[name][1]:.2e} of max|out|"
    assert worst["unit"][0] <= 3e-7                                                         # nothing to fix on well-scaled rows
    assert worst["std 1e-3"][0] >= 3e-6 and worst["mean 50 sigma"][0] >= 1e-6               # the losses the prediction removes
    assert not np.isfinite(worst["std 1e3, mean 50 sigma"][0])                              # fp16 overflow of the unscaled hi plane
