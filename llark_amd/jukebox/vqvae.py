"""Host-side driver of the Jukebox VQ-VAE level-2 encoder on the HIP kernels.

Mirrors what the reference reaches through ``vqvae.encode(x)`` (jukebox/main.py:61; upstream
``VQVAE.encode`` -> ``_encode`` -> ``encoders[level](x)[-1]`` -> ``bottleneck.encode``).  The
reference runs all three level encoders and keeps ``zs[-1]`` (jukebox/main.py:63); the level-2
codes do not depend on levels 0/1, so only the level-2 encoder is executed here and ``encode``
returns ``[None, None, z_top]`` to keep the ``zs[-1]`` access pattern.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import ops
from .hparams import JukeboxHParams

# Near-tie certificate of the fused encoder (csrc/vqvae.hip: llark_codebook_argmin_tie).  A token is re-evaluated exactly when its
# best / second-best codebook gap is below  |x| (4 TIE_E_REL sqrt(d_best) + TIE_ULPS 2^-23 |x|).  What the two terms cover, measured
# over 81 920 tokens of 10 full-size clips (profiles/r04_vq_near_tie_stats.txt; |x| ~ 72, sqrt(d_best) ~ 25, gaps ~ 1e-2 .. 1e2):
#   * the fused stages' output error e = x_fused - x_exact (fp32 accumulation-order noise): |e| / |x| median 1.06e-6, max 2.07e-6;
#     it moves the gap of a code pair by 2 e.(k_b - k_a): rms 4.4e-4, max 2.8e-3 (Cauchy-Schwarz, 4 |e| sqrt(d), would allow
#     1.5e-2: e is not aligned with k_b - k_a);
#   * the fp32 rounding of the distance chain (xx - 2 dot) + kk at |x|^2 ~ 6000 (ulp 4.9e-4), in the oracle's evaluation AND in this
#     one: rms 1.5e-3 per distance, max 6.9e-3 -> 3.0e-3 rms on the difference of two gaps.  THIS is the larger term: every one of
#     the 4 codes (of 65 536) the bare fused argmin gets wrong sits at an exact gap <= 1.3e-3.
# Together sigma = 3.0e-3; the defaults put the threshold at 2.0e-2 = 6.7 sigma (4 TIE_E_REL |x| sqrt(d) = 1.17e-2 = 4.2x the largest
# movement observed, TIE_ULPS ulps = 0.86e-2): ~50 of 65 536 tokens are re-evaluated, and with ~3300 tokens per unit of gap the
# expected number of unresolved flips is ~1e-10 per 8-clip batch.  Wider is safer and slower (15 us per token).
TIE_E_REL = 1.5e-6
TIE_ULPS = 12.0
TIE_LIST_CAP = 4096          # flagged tokens per encode_top call the device list can hold; beyond it the whole batch goes exact
TIE_CHUNK = 256              # windows re-evaluated per pass (bounds the scratch: 256 windows x 32 channels x 11264 positions fp32)


def receptive_halo_tokens(hps: JukeboxHParams) -> int:
    """Tokens either side of a level-2 token whose audio its encoder output depends on: walks upstream's EncoderConvBlock stack
    (vqvae/encdec.py: per level block `down_t` x [Conv1d(k = 2 s, stride s, pad s / 2); Resnet1D(depth, dilations growth^r)] and a
    Conv1d(k = 3, pad 1)) backwards from one output position.  5b: 10455 samples = 82 tokens either side."""
    left = right = 0
    res = sum(hps.dilation_growth_rate ** r for r in range(hps.depth))
    for lb in reversed(range(len(hps.downs_t))):
        left, right = left + 1, right + 1
        s = hps.strides_t[lb]
        for _ in range(hps.downs_t[lb]):
            left, right = left + res, right + res
            left, right = s * left + s // 2, s * right + (2 * s - 1 - s // 2)
    r2t = hps.raw_to_tokens
    return max(-(-left // r2t), -(-max(0, right - (r2t - 1)) // r2t))


class VQVAE:
    """Level-2 encoder + codebook of the Jukebox VQ-VAE, weights resident in HBM in kernel layout."""

    def __init__(self, hps: JukeboxHParams, weights: Dict[str, torch.Tensor], device="cuda", exact: bool = False,
                 tie_e_rel: Optional[float] = TIE_E_REL, tie_ulps: float = TIE_ULPS):
        """exact=False (default): the fused stage kernels on the 16-bit matrix cores (csrc/vqvae_fused.hip: one launch per
        down-sampling step, activations resident in LDS, split-fp16 products -- fp32-class activations) followed by the near-tie
        certificate: every token whose two nearest codebook entries are closer than the threshold below is re-evaluated by the exact
        kernels on a window covering its receptive field.  The threshold is STATISTICAL, not a proof: it sits at 6.7 sigma of the measured
        gap noise and at 4.2x the largest gap movement seen on 81 920 tokens (module header; its Cauchy-Schwarz term uses TIE_E_REL =
        1.5e-6 although single |e| / |x| reach 2.07e-6 -- e is not aligned with k_b - k_a), i.e. the codes equal the exact path's with an
        expected 1e-10 unresolved flips per 8-clip batch; every audited clip so far (10 full clips + the 8 bench clips per run) has 0.
        tie_e_rel=None switches the certificate off (round 3's behaviour: codes may differ on near-ties).
        exact=True: the per-layer fp32 kernels of csrc/vqvae.hip whose activations are BIT-equal to the defined-order C
        oracle (oracle/jukebox_ref.c); ``encoder_forward(..., taps=)`` and ``encode_top(want_dist=True)`` always use them."""
        hps.check()
        self.hps = hps
        self.device = torch.device(device)
        self.sample_length = hps.sample_length
        self.exact = bool(exact)
        self.tie_e_rel = None if tie_e_rel is None else float(tie_e_rel)
        self.tie_ulps = float(tie_ulps)
        self.halo_tokens = receptive_halo_tokens(hps) + 1
        self.win_tokens = min(hps.n_ctx, -(-(2 * self.halo_tokens + 1) // 8) * 8)
        self.last_near_ties = 0                # tokens the last encode_top call re-evaluated exactly
        self.near_ties_total = 0
        self.layers: List[tuple] = []          # ("conv", wp, b, stride, pad) | ("res", w1p, b1, w2p, b2, dil)
        self.stages: List[dict] = []           # fused path: one entry per down-sampling step (see _make_stage)
        dev = self.device

        def f32(name):
            return weights[name].detach().to(device=dev, dtype=torch.float32).contiguous()

        p = "encoders.2"
        for lb, (down_t, stride_t) in enumerate(zip(hps.downs_t, hps.strides_t)):
            for i in range(down_t):
                b = f"{p}.level_blocks.{lb}.model.{i}"
                self.layers.append(("conv", ops.pack_conv_weight(f32(f"{b}.0.weight")), f32(f"{b}.0.bias"),
                                    stride_t, stride_t // 2))
                for r in range(hps.depth):
                    rb = f"{b}.1.model.{r}.model"
                    self.layers.append(("res", ops.pack_conv_weight(f32(f"{rb}.1.weight")), f32(f"{rb}.1.bias"),
                                        ops.pack_conv_weight(f32(f"{rb}.3.weight")), f32(f"{rb}.3.bias"),
                                        hps.dilation_growth_rate ** r))
            b = f"{p}.level_blocks.{lb}.model.{down_t}"
            self.layers.append(("conv", ops.pack_conv_weight(f32(f"{b}.weight")), f32(f"{b}.bias"), 1, 1))
            for i in range(down_t):
                assert stride_t == 2, "the fused stage kernel is built for stride-2 down-sampling (Jukebox strides_t = (2, 2, 2))"
                self.stages.append(self._make_stage(f32, f"{p}.level_blocks.{lb}.model", i, hps, out_conv=(i == down_t - 1),
                                                    down_t=down_t))
        self._plan_handle = None
        self.set_codebook(weights["bottleneck.level_blocks.2.k"])
        self._bufs: Dict[tuple, torch.Tensor] = {}

    @staticmethod
    def _make_stage(f32, prefix: str, i: int, hps: JukeboxHParams, out_conv: bool, down_t: int) -> dict:
        """Packed operands of one fused stage (csrc/vqvae_fused.hip): strided conv, `depth` residual blocks, optional output conv."""
        b = f"{prefix}.{i}"
        w0 = f32(f"{b}.0.weight")                                  # [32][cin][4]
        cin = w0.shape[1]
        st = dict(cin=cin, b0=f32(f"{b}.0.bias"), w0f=None, w0_hi=None, w0_lo=None, wo_hi=None, wo_lo=None, bo=None,
                  dil=[hps.dilation_growth_rate ** r for r in range(hps.depth)])
        wexp = [0]
        if cin == 1:
            st["w0f"] = ops.pack_conv_weight(w0)                  # [4][1][32]: exact fp32 fmaf chain on the VALU
        else:
            wexp[0] = ops.vqvae_weight_exponent(w0)
            st["w0_hi"], st["w0_lo"] = ops.vqvae_pack_frag16(w0, exp2=wexp[0])
        hi, lo, br = [], [], []
        for r in range(hps.depth):
            rb = f"{b}.1.model.{r}.model"
            w3, w1 = f32(f"{rb}.1.weight"), f32(f"{rb}.3.weight")
            e3, e1 = ops.vqvae_weight_exponent(w3), ops.vqvae_weight_exponent(w1)
            h3, l3 = ops.vqvae_pack_frag16(w3, exp2=e3)                            # dilated k = 3 conv: 6 k-steps
            h1, l1 = ops.vqvae_pack_frag16(w1, perm1x1=True, exp2=e1)              # 1x1 conv in accumulator-register channel order
            hi += [h3, h1]
            lo += [l3, l1]
            br += [f32(f"{rb}.1.bias"), f32(f"{rb}.3.bias")]
            wexp += [e3, e1]
        st["wr_hi"], st["wr_lo"], st["br"] = torch.cat(hi).contiguous(), torch.cat(lo).contiguous(), torch.cat(br).contiguous()
        wexp.append(0)
        if out_conv:
            wo = f32(f"{prefix}.{down_t}.weight")                  # [64][32][3]
            wexp[-1] = ops.vqvae_weight_exponent(wo)
            st["wo_hi"], st["wo_lo"] = ops.vqvae_pack_frag16(wo, exp2=wexp[-1])
            st["bo"] = f32(f"{prefix}.{down_t}.bias")
        st["wexp"] = wexp
        return st

    def _planes(self, slot: int, numel: int):
        """Ping-pong (hi, lo) fp16 plane buffers of the fused path."""
        key = ("planes", slot)
        cur = self._bufs.get(key)
        if cur is None or cur[0].numel() < numel:
            cur = (torch.empty((numel,), dtype=torch.float16, device=self.device), torch.empty((numel,), dtype=torch.float16, device=self.device))
            self._bufs[key] = cur
        return cur

    def encoder_forward_fused(self, audio: torch.Tensor) -> torch.Tensor:
        """audio (N, T) fp32 on device -> (N, emb_width, T / raw_to_tokens) fp32, one launch per down-sampling step."""
        n, t = audio.shape
        x, cin, slot = audio.contiguous(), 1, 0
        out = None
        for si, st in enumerate(self.stages):
            assert st["cin"] == cin, f"stage {si}: expects {st['cin']} input channels, has {cin}"
            c = 64 if st["wo_hi"] is not None else 32
            last = si + 1 == len(self.stages)
            if last:
                out = torch.empty((n, c, t // 2), dtype=torch.float32, device=self.device)
                ops.vqvae_stage(x, n, cin, t, st, out_f32=out)
            else:
                planes = self._planes(slot, n * (t // 2) * c)
                ops.vqvae_stage(x, n, cin, t, st, out_planes=planes)
                x, slot = planes, slot ^ 1
            cin, t = c, t // 2
        return out

    def set_codebook(self, k: torch.Tensor) -> None:
        self.k = k.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.kk = ops.codebook_norms(self.k)

    def _buf(self, slot: int, shape) -> torch.Tensor:
        """Two ping-pong activation buffers sized for the widest activation (n x 32 x T/2)."""
        numel = 1
        for s in shape:
            numel *= s
        key = slot
        cur = self._bufs.get(key)
        if cur is None or cur.numel() < numel:
            cur = torch.empty((numel,), dtype=torch.float32, device=self.device)
            self._bufs[key] = cur
        return cur[:numel].view(*shape)

    def encoder_forward(self, x: torch.Tensor, taps: Optional[list] = None) -> torch.Tensor:
        """x: (N, 1, T) fp32 on device -> (N, emb_width, T / raw_to_tokens)."""
        assert x.dim() == 3 and x.shape[1] == 1
        slot = 0
        for layer in self.layers:
            n, c, t = x.shape
            if layer[0] == "conv":
                _, wp, b, stride, pad = layer
                k, cin, cout = wp.shape
                tout = (t + 2 * pad - (k - 1) - 1) // stride + 1
                y = self._buf(slot, (n, cout, tout))
                ops.conv1d(x, wp, b, stride, pad, 1, out=y)
            else:
                _, w1p, b1, w2p, b2, dil = layer
                y = self._buf(slot, (n, c, t))
                ops.resblock(x, w1p, b1, w2p, b2, dil, out=y)
            if taps is not None:
                taps.append(y.clone())
            x = y
            slot ^= 1
        return x

    def _plan(self):
        """C-side layer list for llark_vqvae_encode (built lazily; the packed weights stay owned by self.layers)."""
        if self._plan_handle is None:
            L = ops._lib.lib()
            h = L.llark_vqvae_plan_create()
            if not h:
                raise ops._lib.LlarkHipError("vqvae_plan_create failed")
            for layer in self.layers:
                if layer[0] == "conv":
                    _, wp, b, stride, pad = layer
                    k, cin, cout = wp.shape
                    ops.check(L.llark_vqvae_plan_add_conv(h, wp.data_ptr(), b.data_ptr(), cin, cout, k, stride, pad), "vqvae_plan_add_conv")
                else:
                    _, w1p, b1, w2p, b2, dil = layer
                    ops.check(L.llark_vqvae_plan_add_resblock(h, w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(),
                                                             w1p.shape[2], dil), "vqvae_plan_add_resblock")
            self._plan_handle = h
        return self._plan_handle

    def __del__(self):
        h = getattr(self, "_plan_handle", None)
        if h:
            try:
                ops._lib.lib().llark_vqvae_plan_destroy(h)
            except Exception:
                pass

    def encode_top(self, audio: torch.Tensor, want_dist: bool = False):
        """audio: (N, sample_length) fp32 device tensor -> codes (N, n_ctx) int64.  One C call runs the whole layer list
        (`llark_vqvae_encode`): the ~40 short launches are issued without returning to Python in between."""
        assert audio.dim() == 2 and audio.shape[1] == self.sample_length, (
            f"expected (N,{self.sample_length}) audio, got {tuple(audio.shape)}")
        if want_dist:                                          # per-layer exact path: also returns the distances
            xe = self.encoder_forward(audio.contiguous().view(audio.shape[0], 1, -1))
            return ops.codebook_argmin(xe, self.k, self.kk, want_dist=want_dist)
        n = audio.shape[0]
        audio = audio.contiguous()
        if not self.exact:
            # one HIP-event pair around the whole stack when bench.py times kernels: work = ALGORITHMIC bytes of the stack
            with ops._timed("vqvae_encode", float(self.algorithmic_bytes(n))):
                xe = self.encoder_forward_fused(audio)
                if self.tie_e_rel is None:
                    codes = ops.codebook_argmin(xe, self.k, self.kk)
                else:
                    codes = self._argmin_certified(audio, xe)
            if codes is None:                                   # more near-ties than the list holds: the whole batch goes exact
                return self._encode_top_exact(audio)
            if codes.shape[1] != self.hps.n_ctx:
                raise ops._lib.LlarkHipError(f"vqvae_encode produced {codes.shape[1]} tokens per clip, hparams say {self.hps.n_ctx}")
            return codes
        return self._encode_top_exact(audio)

    def _argmin_certified(self, audio: torch.Tensor, xe: torch.Tensor):
        """Codebook search on the fused encoder output + exact re-evaluation of the near-ties (the one host read of the encoder:
        a 4-byte count).  Returns None when the flag list overflowed."""
        if not hasattr(self, "_flag_count"):
            self._flag_count = torch.zeros((1,), dtype=torch.int32, device=self.device)
            self._flag_list = torch.empty((TIE_LIST_CAP,), dtype=torch.int32, device=self.device)
        codes = ops.codebook_argmin_tie(xe, self.k, self.kk, 4.0 * self.tie_e_rel, self.tie_ulps * 2.0 ** -23, self._flag_count, self._flag_list)
        count = int(self._flag_count.item())
        self.last_near_ties = count
        self.near_ties_total += count
        if count > TIE_LIST_CAP:
            return None
        r2t = self.hps.raw_to_tokens
        wlen = self.win_tokens * r2t
        for i0 in range(0, count, TIE_CHUNK):
            c = min(TIE_CHUNK, count - i0)
            width = max(layer[1].shape[2] for layer in self.layers)
            b0, b1 = self._buf(0, (c * width * (wlen // 2 + 1),)), self._buf(1, (c * width * (wlen // 2 + 1),))
            win = self._buf("win", (c * wlen,))
            col = self._bufs.get("col")
            if col is None:
                col = self._bufs["col"] = torch.empty((TIE_CHUNK,), dtype=torch.int32, device=self.device)
            ops.vqvae_fix_near_ties(self._plan(), audio, r2t, self._flag_list[i0:i0 + c], c, self.halo_tokens, self.win_tokens, win, col,
                                    b0, b1, self.k, self.kk, codes)
        return codes

    def _encode_top_exact(self, audio: torch.Tensor):
        n = audio.shape[0]
        widest = n * max(layer[1].shape[2] for layer in self.layers) * (self.sample_length // 2 + 1)
        b0, b1 = self._buf(0, (widest,)), self._buf(1, (widest,))
        codes = torch.empty((n, self.hps.n_ctx), dtype=torch.int64, device=self.device)
        import ctypes

        t_out = ctypes.c_int(0)
        with ops._timed("vqvae_encode", float(self.algorithmic_bytes(n))):
            ops.check(ops._lib.lib().llark_vqvae_encode(self._plan(), ops._dev(audio, "audio", torch.float32), n, self.sample_length,
                                                        b0.data_ptr(), b1.data_ptr(), widest, ops._dev(self.k, "k", torch.float32),
                                                        ops._dev(self.kk, "kk", torch.float32), self.k.shape[0], codes.data_ptr(),
                                                        ctypes.byref(t_out), ops._stream()), "vqvae_encode")
        if t_out.value != self.hps.n_ctx:
            raise ops._lib.LlarkHipError(f"vqvae_encode produced {t_out.value} tokens per clip, hparams say {self.hps.n_ctx}")
        return codes

    def algorithmic_bytes(self, n: int) -> int:
        """HBM bytes of the level-2 encoder + codebook for n clips with every layer reading its input once and writing its
        output once ("per-layer-fused", 1.42 GB per 1 048 576-sample clip): the numerator of bench.py's
        ``roofline_conv``."""
        total, c, t = 0, 1, self.sample_length
        for layer in self.layers:
            if layer[0] == "conv":
                _, wp, _b, stride, pad = layer
                k, cin, cout = wp.shape
                tout = (t + 2 * pad - (k - 1) - 1) // stride + 1
                total += 4 * (cin * t + cout * tout)
                c, t = cout, tout
            else:
                total += 4 * 2 * c * t
        total += 4 * c * t + 8 * t                      # codebook: encoder output in, int64 codes out
        return n * total + self.k.numel() * 4

    def encode(self, x: torch.Tensor):
        """Upstream-shaped entry: x (N, T, 1) like ``vqvae.encode(torch.cuda.FloatTensor(audio[None,:,None]))``."""
        assert x.dim() == 3 and x.shape[2] == 1
        x = x.to(device=self.device, dtype=torch.float32)
        z = self.encode_top(x[:, :, 0].contiguous())
        return [None, None, z]
