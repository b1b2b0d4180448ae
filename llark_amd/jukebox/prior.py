"""Host-side driver of the Jukebox top prior (SimplePrior / ConditionalAutoregressive2D in
``only_encode`` mode) on the HIP kernels.

Mirrors the objects the reference touches in jukebox/main.py:71-110:
``top_prior.raw_to_tokens``, ``top_prior.labeller.get_batch_labels``, ``top_prior.get_y``,
``top_prior.get_cond`` and ``top_prior.prior.forward(x, x_cond=, y_cond=, encoder_kv=None,
fp16=False)`` with ``top_prior.prior.only_encode = True``.

Data layout in HBM (per batch of N clips, M = N*n_ctx rows):
  h        fp32 [M][W]                      residual stream (updated in place by the GEMM epilogues)
  ln_hi/lo fp16 [M][W]                      LayerNorm output as hi/lo split planes
  qkv      fp32 [M][3S]
  att_hi/lo fp16 [M][Sp]   (Sp = S rounded up to 32, pad columns stay zero)
  g_hi/lo  fp16 [M][Mp]                     QuickGELU output
  ln_part  fp32 [M][2*ceil(W/256)][2]       folded LayerNorm: per-slice (sum, sum of squares) written by the producing epilogue
  ln_stat  fp32 [M][2]                      folded LayerNorm: (mean, rstd) of the row; ln_hi/lo then hold h . gamma, not LN(h)
  weights  fp16 [N_out][Kp]                 transposed (K-contiguous) copies of upstream Conv1D.w
In the opt-in "lo8" precision the three lo planes are E4M3 byte planes (uint8 [M][K rounded up to 64], MFMA slot order, scale 2^12).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import ops
from .hparams import JukeboxHParams


# How the Conv1D products carry the fp32 activation (see PriorTransformer.__init__).  The reference runs them in fp32
# (jukebox/main.py:108, fp16=False): the default is the 22-bit two-pass form, whose 36-layer B = 8 embedding is within the
# ABSOLUTE 1e-4 of BASELINE configs[1] (5.0e-5 measured); "lo8" (15-16 bits, 5.8e-4 absolute = 3e-5 of max|acts|) is opt-in:
# precision="lo8" / LLARK_PRIOR_PRECISION=lo8 / bench.py --prior-precision lo8.
DEFAULT_PRECISION = "f16x2"

# LayerNorm folded into the GEMM epilogues on both sides of it (round 4; include/llark_hip.h, llark_gemm16_ln): the c_proj
# products write hi/lo planes of (x . gamma) and per-slice (sum, sum of squares) next to the residual stream, the c_attn / c_fc
# products apply the row's (mean, rstd) in their epilogues --  LN(x) W = rstd ((x . gamma) W - mean (gamma W)) + (beta W + b) --
# so 71 of the 72 LayerNorm kernels of a 36-layer forward (each one read + one write of the [M][W] stream) disappear.
# LLARK_PRIOR_LN_FOLD=0 (or PriorTransformer(ln_fold=False)) keeps the separate LayerNorm kernel everywhere.
DEFAULT_LN_FOLD = True
# Round 5: the folded planes are pre-normalised with the row's PREDICTED statistics -- the producing
# epilogue writes hi / lo of ((x - shift) scale) . gamma with (shift, scale) = (mean, nearest power of two of rstd) of the same row at its
# previous LayerNorm (llark_gemm16_ln_p / llark_ln_stats_finalize_p; the first prediction of a forward comes from the embedded sequence,
# llark_ln_row_pred), and the finalize kernel hands the consumer ((mean - shift) scale, rstd / scale).  Without it the planes hold x . gamma
# at the residual stream's own level: rows with std ~ 3e-3 (the embedding's scale at layer 0) push the fp16 lo plane into its subnormals
# (2.7e-5 error per product instead of 2e-7, CPU simulation in tests/test_ln_fold_cpu.py), a row mean of 50 sigma costs 6 bits, |x| > 65504 /
# gamma overflows.  LLARK_PRIOR_LN_PRED=0 = round 4's unscaled planes.
DEFAULT_LN_PRED = True
# Round 5 experiment, measured slower and left off: the attention-output product (K = 1216) as the LayerNorm producer on gemm_bda's
# 128x256 tiles, two workgroups to a CU (llark_gemm16_lnp_fragw) instead of the persistent 256x256 tile -- 1.88 ms against 1.78 per
# launch (profiles/r05_cproj_bda_lnp_ab.txt): the co-resident workgroup hides only part of the epilogue and the transposed 32x32
# accumulators reach the fp32 stream in 32-byte row pieces.  LLARK_PRIOR_CPROJ_BDA=1 takes it (same results to rounding).
DEFAULT_CPROJ_BDA = False


class Labeller:
    """Upstream ``Labeller.get_batch_labels`` for the only metadata the reference ever passes
    (artist/genre "unknown" -> id 0; jukebox/main.py:80-91)."""

    def __init__(self, hps: JukeboxHParams):
        self.hps = hps
        self.sample_length = hps.n_ctx * hps.raw_to_tokens

    def get_label(self, artist, genre, lyrics, total_length, offset):
        del lyrics
        if artist != "unknown" or genre != "unknown":
            raise NotImplementedError("only artist/genre 'unknown' (ids 0) are supported; the v3 id tables are "
                                      "not part of the reference tree")
        y = [total_length, offset, self.sample_length, 0, 0] + [-1] * (self.hps.max_bow_genre_size - 1)
        return dict(y=np.array(y, dtype=np.int64))

    def get_batch_labels(self, metas, device="cpu"):
        ys = [self.get_label(**m)["y"] for m in metas]
        return dict(y=torch.from_numpy(np.stack(ys, axis=0)).to(device).long(), info=None)


class _LayerWeights:
    __slots__ = ("ln0_g", "ln0_b", "ln1_g", "ln1_b", "w_attn", "b_attn", "w_proj", "b_proj", "w_fc", "b_fc", "w_proj2",
                 "b_proj2", "sw_attn", "sw_proj", "sw_fc", "sw_proj2", "w8_attn", "w8_proj", "w8_fc", "w8_proj2",
                 "gw_attn", "bw_attn", "gw_fc", "bw_fc", "wf_proj")


class PriorTransformer:
    """``top_prior.prior``: the ConditionalAutoregressive2D forward in ``only_encode`` mode."""

    def __init__(self, hps: JukeboxHParams, weights: Dict[str, torch.Tensor], device, depth: Optional[int] = None,
                 precision: Optional[str] = None, ln_fold: Optional[bool] = None):
        self.hps = hps
        self.device = torch.device(device)
        self.only_encode = False
        # How the Conv1D products carry the fp32 activation (the reference runs them in fp32, jukebox/main.py:108):
        #   "f16x2": fp16 hi + fp16 lo planes, two fp16 MFMA passes (22 significant bits; csrc/gemm256n.hip) -- the default
        #   "lo8"  : fp16 hi + E4M3 lo plane, one fp16 pass + one MX-fp8 MFMA (15-16 bits; csrc/gemm256_lo8n.hip) -- opt-in
        precision = precision or os.environ.get("LLARK_PRIOR_PRECISION", DEFAULT_PRECISION)
        if precision not in ("f16x2", "lo8"):
            raise ValueError(f"prior precision must be 'f16x2' or 'lo8', got {precision!r}")
        self.precision = precision
        if ln_fold is None:
            ln_fold = os.environ.get("LLARK_PRIOR_LN_FOLD", "1" if DEFAULT_LN_FOLD else "0") != "0"
        self.ln_fold = bool(ln_fold) and precision == "f16x2"       # the lo8 tile has no folded epilogues
        self.ln_pred = self.ln_fold and os.environ.get("LLARK_PRIOR_LN_PRED", "1" if DEFAULT_LN_PRED else "0") != "0"
        self.cproj_bda = self.ln_fold and os.environ.get("LLARK_PRIOR_CPROJ_BDA", "1" if DEFAULT_CPROJ_BDA else "0") != "0"
        self.width = hps.prior_width
        self.depth = hps.prior_depth if depth is None else depth
        dev = self.device

        def f32(name):
            return weights[name].detach().to(device=dev, dtype=torch.float32).contiguous()

        def w16(name):
            # upstream Conv1D.w is [n_in][n_out], stored fp16 (fp16_params); pack to [n_out][Kp] fp16
            w = weights[name].detach().to(device=dev)
            if w.dtype != torch.float16:
                w = w.to(torch.float16)      # a real checkpoint stores fp16; fp32 inputs are rounded like upstream
            # (no fragment-major twin here: measured in situ on MI355X, the B-direct GEMM is within +-4 % of the
            #  LDS-staged kernel on the prior's M = 65536 shapes -- both sit at the L2->CU limit -- so the prior
            #  keeps the single weight copy; the Llama engine, M = 2968, gains 8-13 % and attaches one)
            wt = ops.pack_weight16(w.contiguous(), transpose=True, dst_dtype=torch.float16, kmult=64 if precision == "lo8" else 32)
            if precision == "lo8" and wt.shape[1] < 128:          # tiny test configs: zero-pad K to two K-steps
                wt = torch.nn.functional.pad(wt, (0, 128 - wt.shape[1])).contiguous()
            return wt

        self.x_emb = f32("prior.x_emb.weight")
        self.pos_emb = f32("prior.pos_emb.pos_emb")
        self.layers: List[_LayerWeights] = []
        for d in range(self.depth):
            p = f"prior.transformer._attn_mods.{d}"
            L = _LayerWeights()
            L.ln0_g, L.ln0_b = f32(f"{p}.ln_0.weight"), f32(f"{p}.ln_0.bias")
            L.ln1_g, L.ln1_b = f32(f"{p}.ln_1.weight"), f32(f"{p}.ln_1.bias")
            L.w_attn, L.b_attn = w16(f"{p}.attn.c_attn.w"), f32(f"{p}.attn.c_attn.b")
            L.w_proj, L.b_proj = w16(f"{p}.attn.c_proj.w"), f32(f"{p}.attn.c_proj.b")
            L.w_fc, L.b_fc = w16(f"{p}.mlp.c_fc.w"), f32(f"{p}.mlp.c_fc.b")
            L.w_proj2, L.b_proj2 = w16(f"{p}.mlp.c_proj.w"), f32(f"{p}.mlp.c_proj.b")
            if precision == "lo8":      # per-matrix exponent of the in-kernel fp8 weight plane: max|W| * 2^sw <= 448
                L.sw_attn, L.sw_proj = ops.lo8_weight_exponent(L.w_attn), ops.lo8_weight_exponent(L.w_proj)
                L.sw_fc, L.sw_proj2 = ops.lo8_weight_exponent(L.w_fc), ops.lo8_weight_exponent(L.w_proj2)
                # the E4M3 weight planes e4m3(W 2^sw), staged through LDS next to W (+50 % weight bytes: 5.5 GB for the 36 layers)
                L.w8_attn, L.w8_proj = ops.pack_weight_lo8(L.w_attn, L.sw_attn), ops.pack_weight_lo8(L.w_proj, L.sw_proj)
                L.w8_fc, L.w8_proj2 = ops.pack_weight_lo8(L.w_fc, L.sw_fc), ops.pack_weight_lo8(L.w_proj2, L.sw_proj2)
            if self.ln_fold:
                # the folded form's per-column vectors, from the fp16 weights as the kernel reads them, summed in float64:
                #   gw[n] = sum_k gamma_k W[n][k]   (multiplies the row mean),   bw[n] = sum_k beta_k W[n][k] + b[n]
                W = hps.prior_width
                for tag, wt, b, g, be in (("attn", L.w_attn, L.b_attn, L.ln0_g, L.ln0_b), ("fc", L.w_fc, L.b_fc, L.ln1_g, L.ln1_b)):
                    w64 = wt[:, :W].double()
                    setattr(L, "gw_" + tag, (w64 @ g.double()).float().contiguous())
                    setattr(L, "bw_" + tag, (w64 @ be.double() + b.double()).float().contiguous())
                    del w64
            self.layers.append(L)
        self._ws: Dict[str, torch.Tensor] = {}
        self._ws_rows = 0
        self._fold_rows = False

    # ---- workspace ---------------------------------------------------------------------------
    def _workspace(self, rows: int):
        if self._ws_rows != rows:
            hps, dev = self.hps, self.device
            W, S, Mw = hps.prior_width, hps.n_state, hps.mlp_state
            if self.precision == "lo8":       # K-steps of 64, at least two of them (the kernel's double-buffered prologue)
                Sp, Mp, Wp = (max(128, ops.round_up(v, 64)) for v in (S, Mw, W))
            else:
                Sp, Mp, Wp = ops.round_up(S, 32), ops.round_up(Mw, 32), ops.round_up(W, 32)
            lo_dt = torch.uint8 if self.precision == "lo8" else torch.float16       # E4M3 bytes (0x00 = 0.0) / fp16
            ws = {}
            ws["ln_hi"] = torch.zeros((rows, Wp), dtype=torch.float16, device=dev)
            ws["ln_lo"] = torch.zeros((rows, Wp), dtype=lo_dt, device=dev)
            ws["qkv"] = torch.empty((rows, 3 * S), dtype=torch.float32, device=dev)
            ws["att_hi"] = torch.zeros((rows, Sp), dtype=torch.float16, device=dev)   # pad columns stay 0
            ws["att_lo"] = torch.zeros((rows, Sp), dtype=lo_dt, device=dev)
            ws["g_hi"] = torch.zeros((rows, Mp), dtype=torch.float16, device=dev)
            ws["g_lo"] = torch.zeros((rows, Mp), dtype=lo_dt, device=dev)
            # folded LayerNorm: every product of a block has to be a shape the 256x256 tile takes (5b widths at M >= 2048 rows)
            self._fold_rows = bool(self.ln_fold and all(ops.gemm16_ln_takes(rows, nn, kk) for nn, kk in
                                                        ((3 * S, Wp), (W, Sp), (Mw, Wp), (W, Mp))))
            if self._fold_rows:
                self._ln_parts = 2 * ((W + 255) // 256)
                ws["ln_part"] = torch.empty((rows, max(self._ln_parts, (W + 63) // 64 if self.cproj_bda else 0), 2), dtype=torch.float32, device=dev)
                ws["ln_stat"] = torch.empty((rows, 2), dtype=torch.float32, device=dev)
                ws["ln_pred"] = torch.empty((rows, 2), dtype=torch.float32, device=dev) if self.ln_pred else None
            self._ws, self._ws_rows = ws, rows
        return self._ws

    # ---- one ResAttnBlock --------------------------------------------------------------------
    def layer_forward(self, h2: torch.Tensor, d: int, n: int, taps: Optional[dict] = None, fold_in: bool = False,
                      fold_out: Optional[int] = None) -> None:
        """h2: [M][W] fp32 residual stream, updated in place:  a = attn(ln_0(h)); h += a;
        m = mlp(ln_1(h)); h += m   (== upstream ``x + a + m`` evaluated left to right).

        ``fold_in`` / ``fold_out`` chain consecutive blocks in the folded-LayerNorm form (forward() sets them): fold_in = the
        workspace already holds the planes of h . ln_0.gamma and the row statistics (the previous block's fold_out);
        fold_out = index of the block that follows, whose ln_0.gamma this block's last product multiplies in."""
        hps, L = self.hps, self.layers[d]
        rows = h2.shape[0]
        ws = self._workspace(rows)
        W, S, Mw = hps.prior_width, hps.n_state, hps.mlp_state
        if self.precision == "lo8":
            return self._layer_forward_lo8(h2, L, d, n, ws, taps)
        if self._fold_rows and taps is None:
            return self._layer_forward_fold(h2, L, d, n, ws, fold_in, fold_out)
        assert not fold_in, "fold_in without the folded path"
        ops.layernorm_split(h2, L.ln0_g, L.ln0_b, 1e-5, ws["ln_hi"], ws["ln_lo"])
        ops.gemm16(ws["ln_hi"], ws["ln_lo"], L.w_attn, L.b_attn, 3 * S, ops.EPI_F32, c=ws["qkv"])
        ops.prior_attn(ws["qkv"], n, hps.n_ctx, S, hps.heads, hps.blocks, [1, 2, 3][d % 3], ws["att_hi"], ws["att_lo"])
        if taps is not None:
            taps["ln0"] = ws["ln_hi"].float() + ws["ln_lo"].float()
            taps["qkv"] = ws["qkv"].clone()
            taps["att"] = (ws["att_hi"].float() + ws["att_lo"].float())[:, :S]
        ops.gemm16(ws["att_hi"], ws["att_lo"], L.w_proj, L.b_proj, W, ops.EPI_RESID, c=h2, resid=h2)
        if taps is not None:
            taps["xa"] = h2.clone()
        ops.layernorm_split(h2, L.ln1_g, L.ln1_b, 1e-5, ws["ln_hi"], ws["ln_lo"])
        ops.gemm16(ws["ln_hi"], ws["ln_lo"], L.w_fc, L.b_fc, Mw, ops.EPI_QGELU_SPLIT, out_hi=ws["g_hi"], out_lo=ws["g_lo"])
        if taps is not None:
            taps["ln1"] = ws["ln_hi"].float() + ws["ln_lo"].float()
            taps["g"] = (ws["g_hi"].float() + ws["g_lo"].float())[:, :Mw]
        ops.gemm16(ws["g_hi"], ws["g_lo"], L.w_proj2, L.b_proj2, W, ops.EPI_RESID, c=h2, resid=h2)

    def _layer_forward_fold(self, h2, L, d: int, n: int, ws, fold_in: bool, fold_out: Optional[int]) -> None:
        """The block with its LayerNorms folded into the products around them (see DEFAULT_LN_FOLD).  Statistics are per-slice
        sums written by the producing epilogue and reduced in a fixed order: deterministic, batch-size independent."""
        hps = self.hps
        W, S, Mw = hps.prior_width, hps.n_state, hps.mlp_state
        rows, part, stat, pred = h2.shape[0], ws["ln_part"], ws["ln_stat"], ws["ln_pred"]
        if pred is not None and not fold_in:
            ops.ln_row_pred(h2, 1e-5, pred)                  # first block of a chain: predict from the stream as it stands
        if fold_in:
            ops.gemm16_ln(ws["ln_hi"], ws["ln_lo"], L.w_attn, L.bw_attn, 3 * S, ops.EPI_F32, L.gw_attn, ln_stat=stat, c=ws["qkv"])
        else:
            ops.layernorm_split(h2, L.ln0_g, L.ln0_b, 1e-5, ws["ln_hi"], ws["ln_lo"])
            ops.gemm16(ws["ln_hi"], ws["ln_lo"], L.w_attn, L.b_attn, 3 * S, ops.EPI_F32, c=ws["qkv"])
        ops.prior_attn(ws["qkv"], n, hps.n_ctx, S, hps.heads, hps.blocks, [1, 2, 3][d % 3], ws["att_hi"], ws["att_lo"])
        if self.cproj_bda:
            if getattr(L, "wf_proj", None) is None:
                L.wf_proj = ops.pack_weight16_frag(L.w_proj, W)
            np1 = ops.gemm16_lnp_fragw(ws["att_hi"], ws["att_lo"], L.wf_proj, L.b_proj, W, L.w_proj.shape[1], L.ln1_g, part, h2, h2,
                                       ws["ln_hi"], ws["ln_lo"], ln_pred=pred)
        else:
            np1 = self._ln_parts
            ops.gemm16_ln(ws["att_hi"], ws["att_lo"], L.w_proj, L.b_proj, W, ops.EPI_RESID, L.ln1_g, ln_part=part, c=h2, resid=h2,
                          out_hi=ws["ln_hi"], out_lo=ws["ln_lo"], ln_pred=pred)
        ops.ln_stats_finalize(part, rows, np1, W, 1e-5, stat, pred)
        ops.gemm16_ln(ws["ln_hi"], ws["ln_lo"], L.w_fc, L.bw_fc, Mw, ops.EPI_QGELU_SPLIT, L.gw_fc, ln_stat=stat,
                      out_hi=ws["g_hi"], out_lo=ws["g_lo"])
        if fold_out is None:
            ops.gemm16(ws["g_hi"], ws["g_lo"], L.w_proj2, L.b_proj2, W, ops.EPI_RESID, c=h2, resid=h2)
            return
        ops.gemm16_ln(ws["g_hi"], ws["g_lo"], L.w_proj2, L.b_proj2, W, ops.EPI_RESID, self.layers[fold_out].ln0_g, ln_part=part,
                      c=h2, resid=h2, out_hi=ws["ln_hi"], out_lo=ws["ln_lo"], ln_pred=pred)
        ops.ln_stats_finalize(part, rows, self._ln_parts, W, 1e-5, stat, pred)

    def _layer_forward_lo8(self, h2, L, d: int, n: int, ws, taps) -> None:
        """The same block with E4M3 low planes: every producer (LayerNorm, attention, the c_fc epilogue) writes
        fp16(a) + fp8((a - fp16(a)) 2^12), every Conv1D product is one llark_gemm16_lo8."""
        hps = self.hps
        W, S, Mw = hps.prior_width, hps.n_state, hps.mlp_state

        def full(hi, lo8, width):
            return hi.float()[:, :width] + ops.lo8_decode(lo8, width)

        ops.layernorm_split_lo8(h2, L.ln0_g, L.ln0_b, 1e-5, ws["ln_hi"], ws["ln_lo"])
        ops.gemm16_lo8(ws["ln_hi"], ws["ln_lo"], L.w_attn, L.sw_attn, L.b_attn, 3 * S, ops.EPI_F32, c=ws["qkv"], w8=L.w8_attn)
        ops.prior_attn(ws["qkv"], n, hps.n_ctx, S, hps.heads, hps.blocks, [1, 2, 3][d % 3], ws["att_hi"], ws["att_lo"])
        if taps is not None:
            taps["ln0"] = full(ws["ln_hi"], ws["ln_lo"], W)
            taps["qkv"] = ws["qkv"].clone()
            taps["att"] = full(ws["att_hi"], ws["att_lo"], S)
        ops.gemm16_lo8(ws["att_hi"], ws["att_lo"], L.w_proj, L.sw_proj, L.b_proj, W, ops.EPI_RESID, c=h2, resid=h2, w8=L.w8_proj)
        if taps is not None:
            taps["xa"] = h2.clone()
        ops.layernorm_split_lo8(h2, L.ln1_g, L.ln1_b, 1e-5, ws["ln_hi"], ws["ln_lo"])
        ops.gemm16_lo8(ws["ln_hi"], ws["ln_lo"], L.w_fc, L.sw_fc, L.b_fc, Mw, ops.EPI_QGELU_SPLIT8, out_hi=ws["g_hi"], out_lo8=ws["g_lo"], w8=L.w8_fc)
        if taps is not None:
            taps["ln1"] = full(ws["ln_hi"], ws["ln_lo"], W)
            taps["g"] = full(ws["g_hi"], ws["g_lo"], Mw)
        ops.gemm16_lo8(ws["g_hi"], ws["g_lo"], L.w_proj2, L.sw_proj2, L.b_proj2, W, ops.EPI_RESID, c=h2, resid=h2, w8=L.w8_proj2)

    def embed(self, x: torch.Tensor, x_cond: torch.Tensor, y_cond: torch.Tensor) -> torch.Tensor:
        n, t = x.shape
        xc = x_cond.reshape(-1, self.width)[:t].contiguous()
        yc = y_cond.reshape(-1)[: self.width].contiguous()
        return ops.prior_embed(x.contiguous(), self.x_emb, self.pos_emb, xc, yc)

    def forward(self, x, x_cond=None, y_cond=None, encoder_kv=None, fp16=False, depth: Optional[int] = None):
        """``prior.forward(x, x_cond=, y_cond=, encoder_kv=None, fp16=False)`` with only_encode=True
        (jukebox/main.py:105-108).  x: (N, n_ctx) int64 codes.  Returns (N, n_ctx, width) fp32."""
        if not self.only_encode:
            raise NotImplementedError("only the only_encode=True path of the prior is implemented (jukebox/main.py:105)")
        if encoder_kv is not None or fp16:
            raise NotImplementedError("encoder_kv / fp16=True are not used by the reference path (jukebox/main.py:108)")
        assert x_cond is not None and y_cond is not None, "top-level prior is conditioned (x_cond=y_pos, y_cond)"
        x = x.to(self.device)
        n, t = x.shape
        assert t == self.hps.n_ctx, f"expected {self.hps.n_ctx} tokens, got {t}"
        # x_cond / y_cond are the same for every clip (the reference keeps sample 0 only, main.py:95-96)
        h = self.embed(x, x_cond[0:1] if x_cond.dim() == 3 else x_cond, y_cond)
        h2 = h.view(n * t, self.width)
        nd = self.depth if depth is None else depth
        fold = self._workspace(n * t) is not None and self._fold_rows
        for d in range(nd):
            self.layer_forward(h2, d, n, fold_in=fold and d > 0, fold_out=d + 1 if fold and d + 1 < nd else None)
        return h

    __call__ = forward


class TopPrior:
    """``top_prior``: conditioning tables + the transformer (upstream SimplePrior, level = top)."""

    def __init__(self, hps: JukeboxHParams, weights: Dict[str, torch.Tensor], device="cuda", depth: Optional[int] = None,
                 precision: Optional[str] = None, ln_fold: Optional[bool] = None):
        hps.check()
        self.hps = hps
        self.device = torch.device(device)
        self.raw_to_tokens = hps.raw_to_tokens
        self.n_ctx = hps.n_ctx
        self.labeller = Labeller(hps)
        self.prior = PriorTransformer(hps, weights, device, depth, precision, ln_fold)
        dev = self.device
        self._y_emb = {k.split(".")[1]: v.detach().to(device=dev, dtype=torch.float32).contiguous()
                       for k, v in weights.items() if k.startswith("y_emb.")}
        self._cond_cache = None

    def get_y(self, labels, start, get_indices=False):
        y = labels["y"].clone()
        # upstream: y[:, 2] = sample_length ; y[:, 1:2] += start * raw_to_tokens
        y[:, 2] = int(self.n_ctx * self.raw_to_tokens)
        y[:, 1:2] = y[:, 1:2] + int(start * self.raw_to_tokens)
        return y

    def _range_bins(self, n_time: int, rng, pos_start: np.ndarray, pos_end=None, clamp=False) -> np.ndarray:
        """RangeEmbedding bin indices, evaluated in fp32 like upstream (conditioners.py)."""
        pos_min, pos_max = np.float32(rng[0]), np.float32(rng[1])
        pos_start = pos_start.astype(np.float32)
        if pos_end is not None:
            pos_end = pos_end.astype(np.float32)
            if clamp:
                pos_end = np.clip(pos_end, pos_min, pos_max)
        if n_time != 1:
            interpolation = (np.arange(0, n_time, dtype=np.float32).reshape(1, n_time) / np.float32(n_time))
            position = pos_start + (pos_end - pos_start) * interpolation
        else:
            position = pos_start
        normalised = (position - pos_min) / (pos_max - pos_min)
        return np.floor(np.float32(self.hps.t_bins) * normalised.astype(np.float32)).astype(np.int64)

    def get_cond(self, z_conds, y):
        """``SimplePrior.get_cond`` for the top level: x_cond = y_pos, y_cond = start embedding.
        y: (n_samples, 4 + max_bow_genre_size) int64.  Returns (x_cond, y_cond, prime=None).  The
        tables are gathered once (they are constant for the reference's fixed metadata)."""
        assert z_conds is None, "the top-level prior has no upsampler conditioner"
        hps, dev = self.hps, self.device
        yc = y.detach().cpu().numpy().astype(np.int64)
        key = yc.tobytes()
        if self._cond_cache is not None and self._cond_cache[0] == key:
            return self._cond_cache[1]
        n_all = yc.shape[0]
        same = bool((yc == yc[0:1]).all())
        if same:                      # the reference passes n_samples identical rows and keeps [0]
            yc = yc[0:1]
        total_length, offset, length, artist, genre = yc[:, 0:1], yc[:, 1:2], yc[:, 2:3], yc[:, 3:4], yc[:, 4:]
        E = self._y_emb
        artist_emb = E["artist_emb"][torch.from_numpy(artist).to(dev)]                     # (n,1,W)
        mask = torch.from_numpy((genre >= 0).astype(np.float32)).to(dev).unsqueeze(2)
        genre_emb = (E["bow_genre_emb"][torch.from_numpy(np.clip(genre, 0, None)).to(dev)] * mask).sum(dim=1, keepdim=True)
        start_emb = genre_emb + artist_emb
        start, end = offset, offset + length
        tl, st, en = total_length.astype(np.float32), start.astype(np.float32), end.astype(np.float32)
        sr = hps.sr
        r0 = (hps.min_duration * sr, hps.max_duration * sr)
        r1 = (0.0, hps.max_duration * sr)
        r2 = (0.0, 1.0)
        b0 = self._range_bins(1, r0, tl)
        b1 = self._range_bins(hps.n_ctx, r1, st, en)
        b2 = self._range_bins(hps.n_ctx, r2, st / tl, en / tl, clamp=True)
        t = torch.from_numpy
        pos_emb = (E["total_length_emb"][t(b0).to(dev)] + E["absolute_pos_emb"][t(b1).to(dev)]) + E["relative_pos_emb"][t(b2).to(dev)]
        pos_emb, start_emb = pos_emb.contiguous(), start_emb.contiguous()
        if same and n_all > 1:        # zero-copy views instead of n_samples x 157 MB
            pos_emb = pos_emb.expand(n_all, -1, -1)
            start_emb = start_emb.expand(n_all, -1, -1)
        out = (pos_emb, start_emb, None)
        self._cond_cache = (key, out)
        return out
