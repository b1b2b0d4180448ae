"""Probe (not product, not a test): how far is a 16-bit ACTIVATION FLOW from the fp32 oracle at full depth?

Runs oracle/llama_ref.py's forward -- the CPU restatement -- with torch's default device set to the GPU (fp32 matmuls, so only
the rounding points differ) on the 7B x 32-layer fixture of tests/fulldepth.py, once per activation dtype, and compares the
sampled logits rows with the committed fp32 CPU golden (tests/golden/llama7b_full32.npz).  Answers, before any kernel is
written, whether a single-pass fp16 flow (the reference's own GPU dtype: m2t/models/utils.py:129 torch_dtype=float16) would
sit inside north_star's 1e-3 of max|logits|, and what the bf16 flow's figure should be.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fulldepth as FD  # noqa: E402
from oracle import llama_ref as LR  # noqa: E402


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    z = np.load(FD.LLAMA_NPZ)
    spec = FD.llama_spec(32)
    w = {k: v.cuda() for k, v in FD.llama_weights_cpu(spec).items()}
    ids, aud = FD.llama_inputs(1)
    rows = torch.from_numpy(z["rows"]).cuda()
    ref = torch.from_numpy(z["logits_rows"]).cuda().double()
    scale = float(z["logits_maxabs"])
    # weights: are the bf16 values exactly representable in fp16?
    worst_w = 0.0
    for k, v in w.items():
        f = v.float()
        worst_w = max(worst_w, float((f.half().float() - f).abs().max() / f.abs().max()))
    out = {"weights_fp16_repr_err_over_max": worst_w, "logits_max_abs": scale}
    torch.set_default_device("cuda")
    with torch.no_grad():
        for name, dt, rp in (("fp32_gpu", None, False), ("bf16", torch.bfloat16, False), ("fp16", torch.float16, False),
                             ("fp16_round_probs", torch.float16, True)):
            ww = w
            if dt == torch.float16:
                ww = {k: v.float().half() for k, v in w.items()}      # weights as the fp16 matrix cores would see them
            r = LR.forward(ww, spec, ids.cuda(), aud.cuda(), act_dtype=dt, round_probs=rp)
            got = r["logits"][0][rows].double()
            err = float((got - ref).abs().max())
            rms = float((got - ref).pow(2).mean().sqrt())
            agree = float((got.argmax(-1) == ref.argmax(-1)).double().mean())
            hmax = None
            out[name] = {"max_abs_err": err, "err_over_max": err / scale, "rms_err": rms, "argmax_agree": agree}
            print(name, out[name], flush=True)
            del r
        # ---- round 6: 1.5-pass flows.  Every Linear input x is split hi = fp16(x) | bf16(x), lo = x - hi; the hi plane
        # meets the exact 16-bit weight, the lo plane is quantised to E4M3 with one power-of-two scale per 32 contraction elements (OCP MX,
        # what v_mfma_scale_f32_32x32x64_f8f6f4 consumes) and meets an E4M3 copy of the weight (per-tensor or per-32-block scale).
        # q / k / v / probabilities / attention output stay at two-plane (16-bit) precision as in the `split` flow's attention kernel.
        import torch.nn.functional as TF

        def q_e4m3_blocks(t, block=32):
            shp = t.shape
            tb = t.reshape(-1, block)
            amax = tb.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
            e = torch.floor(torch.log2(amax)) - 8.0                    # E4M3: largest binade 2^8 (max normal 448)
            sc = torch.exp2(e)
            qv = (tb / sc).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * sc
            return qv.reshape(shp)

        def q_e4m3_tensor(t):
            amax = float(t.abs().max())
            sw = float(np.floor(np.log2(448.0 / amax))) if amax > 0 else 0.0
            return (t * 2.0 ** sw).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * 2.0 ** (-sw)

        class FShim:
            def __init__(self, hi_dtype, wmode):
                self.hi_dtype, self.wmode, self.cache = hi_dtype, wmode, {}

            def __getattr__(self, name):
                return getattr(TF, name)

            def linear(self, x, wt, b=None):
                if wt.shape[1] % 32 != 0 or wt.shape[0] < 1024:          # (projector K = 4800 is a multiple of 32 too; tiny heads stay exact)
                    return TF.linear(x, wt, b)
                hi = x.to(self.hi_dtype).float()
                lo = q_e4m3_blocks(x - hi)
                key = wt.data_ptr()
                w8 = q_e4m3_blocks(wt) if self.wmode == "mx" else q_e4m3_tensor(wt)
                return TF.linear(hi, wt, b) + TF.linear(lo, w8)

        orig_F, orig_rnd = LR.F, LR._rnd
        for name, hi_dtype, wmode in (("fp16_hi_mxfp8_lo_wtensor", torch.float16, "tensor"), ("fp16_hi_mxfp8_lo_wmx", torch.float16, "mx"),
                                      ("bf16_hi_mxfp8_lo_wmx", torch.bfloat16, "mx")):
            LR.F = FShim(hi_dtype, wmode)
            LR._rnd = lambda x, act_dtype: x
            try:
                r = LR.forward(w, spec, ids.cuda(), aud.cuda(), act_dtype=None, round_probs=False)
            finally:
                LR.F, LR._rnd = orig_F, orig_rnd
            got = r["logits"][0][rows].double()
            err = float((got - ref).abs().max())
            out[name] = {"max_abs_err": err, "err_over_max": err / scale, "rms_err": float((got - ref).pow(2).mean().sqrt()),
                         "argmax_agree": float((got.argmax(-1) == ref.argmax(-1)).double().mean())}
            print(name, out[name], flush=True)
            del r
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "llama_flow_error.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
