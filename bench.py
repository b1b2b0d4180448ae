#!/usr/bin/env python
"""Headline benchmark: clips/sec of LLark's audio-encoder -> LLM forward on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 works both ways: started plainly (no WORLD_SIZE in the environment) it launches its own N ranks, one per GPU,
  under torch.distributed.run (like scripts/training/train_llark.sh:20-22 of the reference); started BY
  ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` it is one of those ranks.

One "step" = one pass of the hot path over one batch of B synthetic clips per GPU
(25 s @ 44.1 kHz audio, truncated to 1 048 576 samples as jukebox/main.py:59 does, + a 128-token
prompt):  VQ-VAE level-2 encode -> 36-layer Jukebox top prior -> 10 fps mean-pool (240 x 4800)
-> mm_projector + splice -> Llama-2-7B causal-LM forward over S = 371 positions -> logits.
Clips are independent units: every rank processes its own B clips (weak scaling, no data-path
collective); the only collectives are the barrier and the max-over-ranks of the elapsed time.
Inputs are resident in HBM when the timed region starts.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)


def _load_fulldepth_llm_parity():
    """Errors of the Llama flows against the fp32 CPU oracle at the benchmarked depth (7B x 32 layers, S = 371, B = 8), as recorded by
    tests/test_fulldepth_gpu.py on a GPU box (profiles/r05_llama_fulldepth_parity.json).  Annotation only: nothing timed reads it."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_llama_fulldepth_parity.json")) as f:
            rec = json.load(f)
        return {k: {"logits_err_over_max": v.get("logits_err_over_max"), "greedy_tokens_matching": v.get("greedy_tokens_matching"),
                    "greedy_tokens": v.get("greedy_tokens"), "source": "tests/test_fulldepth_gpu.py -> profiles/r05_llama_fulldepth_parity.json"}
                for k, v in rec.items()}
    except (OSError, ValueError):
        return {}


FULLDEPTH_LLM_PARITY = _load_fulldepth_llm_parity()
PEAK_HBM_GBS = 8000.0


def _dist_setup(n_gpus: int, backend: str = "nccl"):
    from llark_amd import dist as D

    rank, world, local = D.env_rank_world()
    if world != n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {n_gpus}, "
                         "or start bench.py plainly and it spawns its own ranks)")
    if backend == "nccl":
        if local >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} needs cuda:{local} but only {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(local)                 # one process per GPU
        D.init(backend="nccl", device=torch.device("cuda", local))      # "nccl" IS RCCL on ROCm; only barrier + max-reduce use it
    else:
        D.init(backend=backend)
    return rank, world, local


def _barrier(world, cuda=True):
    from llark_amd import dist as D

    D.barrier(world, cuda=cuda)


def dist_check(args):
    """--stages dist-check: the launcher + rendezvous + barrier + max-over-ranks + gather of bench.py with NO kernel
    work (each rank sleeps rank-dependent milliseconds).  Runs on CPU over gloo (tests/test_dist_cpu.py drives the
    self-launch path through it) and on GPUs over RCCL."""
    from llark_amd import dist as D

    rank, world, local = _dist_setup(args.gpus, args.backend)
    cuda = args.backend == "nccl"
    device = torch.device("cuda", local) if cuda else "cpu"
    _barrier(world, cuda)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    mine = time.perf_counter() - t0
    _barrier(world, cuda)
    elapsed = D.max_over_ranks(time.perf_counter() - t0, world, device=device)
    per_rank = D.gather_floats(mine / args.steps * 1e3, world, device=device)
    if rank == 0:
        import torch.distributed as dist

        print(json.dumps({"metric": "dist-check (no kernel work)", "value": round(args.steps * world / elapsed, 3), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "backend": args.backend,
                          "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                          "clips_by_rank": [D.clip_indices(r, 2) for r in range(world)],
                          "per_rank_ms_per_step": [round(v, 3) for v in per_rank]}), flush=True)
    D.shutdown(world)


def _prior_gemm_flops(hps, rows):
    W, S, M = hps.prior_width, hps.n_state, hps.mlp_state
    per_layer = 2.0 * rows * (W * 3 * S + S * W + W * M + M * W)
    return per_layer * hps.prior_depth


def build_workload(args, device):
    from llark_amd.jukebox import extract as E
    from llark_amd.jukebox.hparams import hparams_5b, hparams_tiny
    from llark_amd.jukebox.synthetic import init_codebook_from_encodings, make_jukebox_weights, synthetic_clip

    hps = hparams_tiny() if args.tiny else hparams_5b()
    if args.stages == "clap":
        from llark_amd.m2t import bench_support

        return hps, None, None, None, bench_support.ClapWorkload(args, device, "fp32" if args.llm_precision == "split" else "bf16")
    if args.stages == "mpt":
        from llark_amd.m2t import bench_support

        args.with_clap = True
        return hps, None, None, None, bench_support.MptWorkload(args, device)
    if args.stages == "mpt-train":
        from llark_amd.m2t import bench_support

        return hps, None, None, None, bench_support.MptTrainWorkload(args, device, int(os.environ.get("WORLD_SIZE", "1")))
    if args.stages == "train":
        from llark_amd.m2t import bench_support

        world = int(os.environ.get("WORLD_SIZE", "1"))
        return hps, None, None, None, bench_support.TrainWorkload(args, device, world, layers=args.llm_layers or None)
    if args.depth:
        hps.prior_depth = args.depth
    weights = make_jukebox_weights(hps, seed=0, device=device)
    enc = E.WrappedAudioEncoder(hps=hps, weights=weights, device=device, precision=args.prior_precision)
    args.prior_precision = enc.top_prior.prior.precision
    args.prior_ln_fold = bool(enc.top_prior.prior.ln_fold)
    seconds = 25.0 if not args.tiny else 1.6
    rank = int(os.environ.get("RANK", "0"))

    def clip(i):
        a = E._normalize(synthetic_clip(i, seconds=seconds))
        a = E.maybe_pad_audio_to_max_len(a, hps.sample_length)[: hps.sample_length]
        return a.astype(np.float32)

    # data-dependent codebook (upstream init_k analogue) from a calibration clip through the HIP encoder
    cal = torch.from_numpy(clip(100000)).to(device)[None, None, :]
    xe = enc.vqvae.encoder_forward(cal)[0]
    enc.vqvae.set_codebook(init_codebook_from_encodings(xe, hps.l_bins))
    weights["bottleneck.level_blocks.2.k"] = enc.vqvae.k.cpu()
    from llark_amd.dist import clip_indices

    audio = torch.from_numpy(np.stack([clip(i) for i in clip_indices(rank, args.batch)])).to(device)
    llm = None
    if args.stages in ("e2e", "llama", "generate"):
        from llark_amd.m2t import bench_support

        llm = bench_support.build(args, device)
    return hps, weights, enc, audio, llm


def _median_time(fn, warm: int, reps: int):
    """BASELINE.md section 2 protocol, bounded: `warm` untimed + `reps` timed calls -> (median, min) seconds."""
    import statistics

    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
    return statistics.median(ts), min(ts)


def cpu_baseline(hps, weights, args):
    """The CPU oracle on the host cores, bounded sample, per stage `cpu_warm` warm-up + `cpu_reps` timed runs, MEDIAN
    (BASELINE.md section 2): 1 clip through the VQ-VAE (bit-exact C restatement, OpenMP) + `cpu_layers` prior layers (torch
    fp32) extrapolated to 36 (+ the Llama legs of cpu_baseline_llm).  Returns the baseline dict; its "parts" (seconds per
    clip and stage) also feed the generate leg."""
    from llark_amd.jukebox.synthetic import synthetic_clip
    from oracle import jukebox_c as C
    from oracle import jukebox_ref as R

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    seconds = 25.0 if not args.tiny else 1.6
    a = R.normalize_audio(synthetic_clip(0, seconds=seconds))
    a = np.pad(a, (0, max(0, hps.sample_length - len(a))))[: hps.sample_length].astype(np.float32)
    wc = {k: v.detach().float().cpu() if not k.endswith(".w") else v.detach().cpu() for k, v in weights.items()
          if not k.startswith("prior.transformer") or int(k.split(".")[3]) < args.cpu_layers}
    box = {}

    def enc():
        box["z"] = torch.from_numpy(C.encode_codes(wc, a[None], hps))

    t_enc, t_enc_min = _median_time(enc, 0, args.cpu_reps)
    x_cond, y_cond = R.get_cond(wc, hps)
    h0 = R.prior_embed(wc, box["z"], x_cond, y_cond, hps)

    def layers():
        h = h0
        for d in range(args.cpu_layers):
            h = R.prior_layer(wc, h, d, hps)

    t_layers, t_layers_min = _median_time(layers, args.cpu_warm, args.cpu_reps)
    t_prior = t_layers / args.cpu_layers * hps.prior_depth
    total = t_enc + t_prior
    proto = f"{args.cpu_warm} warm-up + {args.cpu_reps} timed runs per stage, median (min)"
    sample = (f"1 clip, {proto}: VQ-VAE encode (C oracle) {t_enc:.2f}s ({t_enc_min:.2f}) + {args.cpu_layers} of {hps.prior_depth} prior layers "
              f"(torch fp32) {t_layers:.2f}s ({t_layers_min:.2f}) extrapolated to {hps.prior_depth}")
    parts = {"vqvae_s": t_enc, "prior_s": t_prior}
    if args.stages in ("e2e", "llama", "generate"):
        t_llm, s_llm, llm_parts = cpu_baseline_llm(args)
        total += t_llm
        sample += "; " + s_llm
        parts.update(llm_parts)
    return {"value": 1.0 / total, "unit": "clips/s", "cores": cores, "kind": "port", "sample": sample,
            "parts_s": {k: round(v, 3) for k, v in parts.items()}}


def cpu_baseline_llm(args):
    """Oracle Llama forward on the host: B=1, S=371, `cpu_layers` of 32 layers at full width + lm_head, extrapolated to 32
    layers; then greedy decode steps against the oracle's KV cache (one token, same layer sample) for the generate leg.
    Same warm-up / repeat / median protocol."""
    from llark_amd.m2t import bench_support as BS
    from oracle import llama_ref as LR

    layers = max(1, args.cpu_layers)
    spec = LR.LlamaSpec(num_hidden_layers=layers, vocab_size=BS.VOCAB, audio_start_token=BS.START, audio_end_token=BS.END,
                        audio_patch_token=BS.PATCH)
    w = LR.make_weights(spec, seed=0, std=0.02)
    ids = BS.make_prompt_ids(1)
    aud = torch.randn(1, BS.FRAMES, 4800)
    box = {}

    def head():
        LR.forward(w, spec, ids, aud, num_layers=0)

    def full():
        box["out"] = LR.forward(w, spec, ids, aud)

    t_head, t_head_min = _median_time(head, args.cpu_warm, args.cpu_reps)
    t_all, t_all_min = _median_time(full, args.cpu_warm, args.cpu_reps)
    t_layers = max(t_all - t_head, 1e-6)
    total = t_head + t_layers / layers * 32
    past = box["out"]["past_key_values"]
    nxt = box["out"]["logits"][:, -1].argmax(-1, keepdim=True)

    def dec_head():
        LR.forward(w, spec, nxt, None, num_layers=0)

    def dec():
        LR.forward(w, spec, nxt, None, past_key_values=past)

    t_dh, _ = _median_time(dec_head, 1, args.cpu_reps)
    t_d, t_d_min = _median_time(dec, 1, args.cpu_reps)
    t_tok = t_dh + max(t_d - t_dh, 1e-6) / layers * 32
    return total, (f"Llama fwd B=1 S=371 fp32 oracle: embed+projector+lm_head {t_head:.2f}s ({t_head_min:.2f}) + {layers} of 32 layers "
                   f"{t_layers:.2f}s extrapolated to 32; one cached decode step {t_tok:.2f}s (head {t_dh:.2f}s + {layers} layers {max(t_d - t_dh, 0):.2f}s x 32/{layers})"), \
        {"llama_prefill_s": total, "llama_decode_token_s": t_tok}


def cpu_baseline_train(args, layers=None):
    """Oracle training step on the host (torch fp32 autograd over oracle/llama_ref.py): 1 clip, S = train_seq,
    `cpu_layers` of 32 layers + lm_head, extrapolated to 32 layers (optimizer step excluded).  One timed pass each (a pass is
    tens of seconds at S = 2048: the bounded-sample rule of the bench contract wins over the repeat count here)."""
    import os

    from llark_amd.m2t import bench_support as BS
    from oracle import llama_ref as LR

    layers = max(1, layers or args.cpu_layers)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    spec = LR.LlamaSpec(num_hidden_layers=layers, vocab_size=BS.VOCAB, audio_start_token=BS.START, audio_end_token=BS.END,
                        audio_patch_token=BS.PATCH)
    w = LR.make_weights(spec, seed=0, std=0.02)
    for k, v in w.items():
        if v.is_floating_point() and "lm_head" not in k:
            v.requires_grad_(True)
    ids = BS.make_prompt_ids(1)
    ans = torch.randint(3, 32000, (1, args.train_seq - ids.shape[1]))
    full = torch.cat([ids, ans], 1)
    labels = full.clone()
    labels[:, : ids.shape[1]] = -100
    aud = torch.randn(1, BS.FRAMES, 4800)

    def run(nl):
        t0 = time.time()
        out = LR.forward(w, spec, full, aud, labels=labels, num_layers=nl)
        out["loss"].backward()
        return time.time() - t0

    t_head = run(0)
    t_all = run(None)
    t_layers = max(t_all - t_head, 1e-6)
    total = t_head + t_layers / layers * 32
    return {"value": 1.0 / total, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"1 clip fwd+bwd S={args.train_seq} fp32 torch-autograd oracle: head {t_head:.2f}s + {layers} of 32 layers {t_layers:.2f}s extrapolated to 32"}


PRIOR_DT = {"f16x2": "fp16x2-split(fp32-class)", "lo8": "fp16+e4m3-split(fp32 accumulate, 15-bit activations)"}
# The prior's Conv1D products per precision: HIP-event timer name, kernel, issued matrix work per algorithmic flop in units of the
# fp16 MFMA rate (f16x2 = two fp16 passes; lo8 = one fp16 pass + one MX-fp8 MFMA per 64 k, 64 cycles against the 128 of four fp16
# MFMAs = 1.5), and the committed PMC record of that kernel (separate --pmc passes: FETCH_SIZE x2 on gfx950 + WRITE_SIZE).
PRIOR_GEMM = {
    "f16x2": ("gemm_split_f16", "gemm256x_kernel<f16> (v_mfma_f32_16x16x32_f16: the K = 4800 products, and the K = 1216 product when its epilogue produces the next LayerNorm's operand) / gemm256n_kernel<f16> (K = 1216 without the fold; 32x32x16): two fp16 MFMA passes, fp32 accumulate", 2.0, "r05_pmc_gemm256x.json"),
    "lo8": ("gemm_lo8_f16", "gemm256_lo8n_kernel (fp16 hi pass + MX-fp8 low plane)", 1.5, "r02_pmc_gemm_lo8.json"),
}


def prior_roofline(timers, hps, args, precision):
    """MFMA roofline of the prior's GEMMs from the HIP events of the timed region: ALGORITHMIC flops (2*M*N*K of the
    fp32-equivalent product; the second pass is not counted) / event time of the launches."""
    tname, kernel, passes, pmc_name = PRIOR_GEMM[precision]
    if tname not in timers:
        return None
    folded = bool(getattr(args, "prior_ln_fold", False)) and precision == "f16x2"
    launches, ms, _ = timers[tname]
    flops = _prior_gemm_flops(hps, args.batch * hps.n_ctx) * args.steps
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, traffic_note = None, "no PMC record committed for this kernel version"
    pmc = os.path.join(ROOT, "profiles", pmc_name)
    # the record is of ONE launch shape (M = 8 clips x 8192 rows, 5b widths): attached to that workload only
    if os.path.exists(pmc) and args.batch == 8 and hps.n_ctx == 8192 and hps.prior_width == 4800:
        d = json.load(open(pmc))
        traffic = d["traffic_bytes_per_launch"]
        traffic_note = ("NOT measured in this run: memory-side bytes of ONE launch of %s from the committed rocprofv3 --pmc passes in "
                        "profiles/%s (algorithmic %.2f GB; the counter includes Infinity-Cache hits)"
                        % (d["kernel"], pmc_name, d["algorithmic_bytes_per_launch"] / 1e9))
    return {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
            "traffic_source": ("profiles/" + pmc_name) if traffic is not None else None,
            "launches": launches, "avg_launch_ms": round(ms / launches, 4), "mfma_passes": passes,
            "frac_of_issued_mfma": round(passes * achieved / PEAK_F16_MFMA_TFLOPS, 4),
            "layernorm_folded": folded,
            "layernorm_note": ("the c_proj launches also write the next LayerNorm's operand planes + row statistics and the c_attn / c_fc launches apply "
                               "them (71 of the 72 LayerNorm kernels of a forward are gone: -11 ms per step end to end, measured A/B in round 4); that "
                               "work is inside the event time this fraction is computed from, its flops are not counted "
                               "(LLARK_PRIOR_LN_FOLD=0: 0.27 with the LayerNorm kernels timed separately)") if folded else None}


def conv_roofline(timers, args):
    """HBM roofline of the VQ-VAE level-2 encoder + codebook (north_star: >= 50 %): ALGORITHMIC bytes (every layer reads its input
    once and writes its output once, 1.42 GB per clip -- SURVEY 8(d)) / HIP-event time of the llark_vqvae_encode call."""
    if "vqvae_encode" not in timers:
        return None
    launches, ms, work = timers["vqvae_encode"]
    gbs = work / (ms * 1e-3) / 1e9
    traffic, traffic_note = None, "no PMC record committed"
    try:                                                              # committed record of the same kernels (annotation: nothing timed reads it)
        rec = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_conv.json")))
        traffic = int(rec["traffic_bytes_per_call"] * args.batch / rec["clips_per_call"])
        traffic_note = ("NOT measured in this run: memory-side bytes of one encode_top call (scaled to this batch) from the committed rocprofv3 --pmc passes in "
                        "profiles/r05_pmc_conv.json (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE): %.2fx the per-layer-fused algorithmic bytes -- the fused stages keep "
                        "intermediate activations in LDS" % rec["traffic_over_algorithmic"])
    except (OSError, ValueError, KeyError):
        pass
    return {"bound": "hbm", "kernel": "llark_vqvae_encode (conv / resblock chain + codebook argmin, one event pair per batch)",
            "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
            "algorithmic_gb_per_clip": round(work / launches / args.batch / 1e9, 4), "launches": launches,
            "ms_per_clip": round(ms / launches / args.batch, 4)}


def roofline_clap(timers, args, wl):
    """HTSAT's matrix products on the 16-bit MFMA GEMM: algorithmic 2*M*N*K of one step / HIP-event time of the GEMM launches
    (fp32 mode issues 3 bf16 MFMA passes per product: A_hi.W_hi + A_lo.W_hi + A_hi.W_lo)."""
    from llark_amd.clap import algorithmic_flops_per_clip

    fl = algorithmic_flops_per_clip(wl.dims)
    keys = [k for k in timers if k.startswith("gemm_")]
    if not keys:
        return None
    ms = sum(timers[k][1] for k in keys)
    launches = sum(timers[k][0] for k in keys)
    achieved = fl["gemm"] * args.batch * args.steps / (ms * 1e-3) / 1e12
    passes = 3 if args.llm_precision == "split" else 1
    return {"bound": "mfma", "kernel": "gemm_kernel<bf16> (HTSAT linears)", "achieved": round(achieved, 2), "peak": PEAK_F16_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4), "traffic": None, "launches": launches,
            "avg_launch_ms": round(ms / launches, 4), "mfma_passes": passes,
            "frac_of_issued_mfma": round(passes * achieved / PEAK_F16_MFMA_TFLOPS, 4),
            "note": "stage-0/1 products (K = 128 / 256 over 262144 / 65536 rows) are HBM-bound on their fp32 outputs, not MFMA-bound"}


def cpu_baseline_clap(wl):
    """The CPU oracle (numpy float64 log-mel + torch fp32 HTSAT) on 2 of the step's clips, same synthetic weights.
    Thread count capped at 16: the oracle's products are small and slow down beyond that on a many-core host."""
    from oracle import clap_ref as CR

    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    from llark_amd.clap import random_state_dict

    w = {k: v.cpu() for k, v in random_state_dict(wl.dims, device=wl.wav.device, seed=0).items()}
    wav = wl.wav[:2].cpu().numpy()
    t0 = time.time()
    feats = np.stack([CR.logmel(CR.quantize_roundtrip(x)) for x in wav]).astype(np.float32)
    with torch.no_grad():
        CR.forward(w, CR.ClapSpec(), torch.from_numpy(feats)[:, None])
    dt = time.time() - t0
    return {"value": 2.0 / dt, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"2 clips: float64 numpy log-mel + torch fp32 HTSAT-base oracle ({dt:.2f}s)"}


def extra_generate(args, enc, audio, llm, cpu, steps: int = 3, new_tokens: int = 64):
    """configs[2] attached to the default line: ONE clip -> Jukebox embed -> projector -> Llama-2-7B prefill (S = 371) + 64 greedy
    decode steps against the KV cache (m2t/generate.py path; stopping criterion off), `steps` timed repeats after one warm-up.
    roofline: the decode steps against HBM (every weight byte once per token)."""
    with torch.no_grad():
        a1 = audio[:1].contiguous()
        llm.generate(enc(a1), new_tokens, batch=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec_ms = 0.0
        for _ in range(steps):
            llm.generate(enc(a1), new_tokens, batch=1, time_decode=True)
            torch.cuda.synchronize()
            dec_ms += llm.decode_events[0].elapsed_time(llm.decode_events[1])
        dt = (time.perf_counter() - t0) / steps
    per_tok = dec_ms / steps / (new_tokens - 1)
    gbs = llm.decode_weight_bytes() / (per_tok * 1e-3) / 1e9
    out = {"metric": "clips/sec embed + prefill + %d-token greedy decode (B = 1)" % new_tokens, "value": round(1.0 / dt, 4), "unit": "clips/s",
           "ms_per_clip": round(dt * 1e3, 2), "steps": steps, "warmup": 1, "decode_ms_per_token": round(per_tok, 4),
           "config": {"workload": "configs[2]: 1 clip -> Jukebox embed -> projector -> Llama-2-7B prefill (S=371) + %d greedy decode steps" % new_tokens,
                      "llm_precision": args.llm_precision, "prior_precision": args.prior_precision,
                      "prior_ln_fold": getattr(args, "prior_ln_fold", None)},
           "roofline": {"bound": "hbm", "kernel": "decode step (weight-streaming Linears: llark_gemv16_dma / gemm_skinny_kernel)", "achieved": round(gbs, 1),
                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                        "bytes_per_token": llm.decode_weight_bytes(), "note": "algorithmic bytes = every bf16 weight once per token; time = HIP events around the %d decode steps" % (new_tokens - 1)}}
    if cpu is not None and "parts_s" in cpu and "llama_decode_token_s" in cpu["parts_s"]:
        ps = cpu["parts_s"]
        tot = ps["vqvae_s"] + ps["prior_s"] + ps["llama_prefill_s"] + new_tokens * ps["llama_decode_token_s"]
        out["cpu_baseline"] = {"value": 1.0 / tot, "unit": "clips/s", "cores": cpu["cores"], "kind": "port",
                               "sample": "the e2e line's CPU legs (embed + prefill) + %d x one cached oracle decode step (%.2f s, extrapolated from %d layers)"
                                         % (new_tokens, ps["llama_decode_token_s"], args.cpu_layers)}
    return out


def extra_train(args, device, steps: int = 2):
    """configs[3] per-GPU share at the reference recipe's micro-batch attached to the default line: 8 clips x 2048 tokens = micro-batch 2 x
    accumulation 4 (scripts/training/train_llark.sh:25-27,40), fwd + bwd + grad-norm clip + AdamW, one warm-up + `steps` timed."""
    import copy

    from llark_amd import dist as D
    from llark_amd.m2t import bench_support

    a = copy.copy(args)
    a.batch, a.micro_batch, a.train_seq, a.grad_comm, a.grad_checkpoint = 8, 2, 2048, "bf16", False
    torch.cuda.reset_peak_memory_stats()
    wl = bench_support.TrainWorkload(a, device, 1)
    wl.trainer.time_phases = True
    with torch.no_grad():
        wl.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ev = wl.trainer.phase_events
        fwd_ms, bwd_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    d = wl.dims
    layer_bytes = (4 * d.hidden_size * d.hidden_size + 3 * d.hidden_size * d.intermediate_size) * 2.0          # bf16 on the links
    rest_bytes = (d.hidden_size * d.mm_hidden_size + 2 * d.hidden_size * 33) * 2.0
    model = {}
    for name, links in (("one_ring", 1), ("all_7_links", D.XGMI_LINKS_PER_GPU)):
        tot, exposed = D.model_overlapped_exchange(bwd_ms, d.num_hidden_layers, layer_bytes, rest_bytes, 8, links)
        model[name] = {"allreduce_ms_at_8": round(tot, 1), "exposed_ms": round(exposed, 1)}
    out = {"metric": "clips/sec instruction-tuning step (fwd+bwd+grad-norm clip+AdamW), per GPU", "value": round(a.batch / dt, 4), "unit": "clips/s",
           "ms_per_step": round(dt * 1e3, 2), "steps": steps, "warmup": 1, "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1),
           "config": {"workload": "configs[3] per-GPU share: 8 clips x 2048 tokens = micro-batch 2 x accumulation 4, random-init Llama-2-7B + projector on "
                                  "frozen features; bf16 (fp32 accumulate, fp32 grads + AdamW moments)", "n_gpus": 1},
           "mfu": round(wl.model_flops_per_step() / dt / (PEAK_F16_MFMA_TFLOPS * 1e12), 4),
           "roofline": {"bound": "mfma", "kernel": "whole step (6 x layer params x tokens + 4 x V x H x tokens)", "achieved": round(wl.model_flops_per_step() / dt / 1e12, 1),
                        "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(wl.model_flops_per_step() / dt / (PEAK_F16_MFMA_TFLOPS * 1e12), 4), "traffic": None},
           "last_micro_batch_ms": {"forward": round(fwd_ms, 2), "backward": round(bwd_ms, 2)},
           "allreduce_model_at_8_gpus": {"bytes_bf16": layer_bytes * d.num_hidden_layers + rest_bytes,
                                         "assumed": "MODEL, not a measurement: ring/direct all-reduce moves 2*(N-1)/N of the buffer per GPU over xGMI links of %.0f GB/s; per-layer slices "
                                                    "issued as the measured backward of the LAST micro-batch completes them (llark_amd.dist.model_overlapped_exchange)" % D.XGMI_LINK_GBS,
                                         **model}}
    if not args.no_cpu_baseline:
        a.cpu_layers = 1
        out["cpu_baseline"] = cpu_baseline_train(a, layers=1)
    del wl
    torch.cuda.empty_cache()
    # The same optimizer step with the accumulation FUSED: the 4 micro-batches as one pass of 8 clips, the loss normalised per micro-batch
    # (HipLlamaTrainer.forward_backward(loss_groups=4): the same gradient up to the order of fp32 sums, tests/test_train_gpu.py) -- every dW
    # product writes its fp32 gradient once instead of read-modify-writing it per micro-batch; activations of 8 clips instead of 2 in HBM.
    try:                                                    # (its own guard: a failure here must not cost the recipe-shaped record above)
        a.micro_batch, a.loss_groups = 8, 4
        torch.cuda.reset_peak_memory_stats()
        wl = bench_support.TrainWorkload(a, device, 1)
        with torch.no_grad():
            wl.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                wl.step()
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - t0) / steps
        out["fused_accumulation"] = {"what": "the same 8 clips x 2048 tokens as ONE pass, loss normalised per recipe micro-batch (loss_groups=4): same optimizer step",
                                     "ms_per_step": round(dtf * 1e3, 2), "value": round(a.batch / dtf, 4), "unit": "clips/s",
                                     "mfu": round(wl.model_flops_per_step() / dtf / (PEAK_F16_MFMA_TFLOPS * 1e12), 4),
                                     "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
        del wl
    except Exception as e:  # noqa: BLE001
        out["fused_accumulation"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return out


def extra_mpt(args, device, steps: int = 3, new_tokens: int = 64):
    """configs[4] per-GPU share attached to the default line: 8 x 10 s 48 kHz clips -> log-mel -> CLAP HTSAT-base -> (1, 512) embedding ->
    projector -> MPT-1B prefill (S = 132) + 64 greedy decode steps (scripts/clap/clap_embeddings.py:63-153 + m2t/models/mpt.py), and the
    audio half alone at batch 64.  No CPU leg here (the HTSAT oracle takes ~40 s per clip on the host: `--stages clap` times it)."""
    import copy

    from llark_amd import ops
    from llark_amd.m2t import bench_support

    a = copy.copy(args)
    a.batch, a.with_clap, a.new_tokens = 8, True, new_tokens
    out = {}
    with torch.no_grad():
        wl = bench_support.MptWorkload(a, device)
        wl.generate(new_tokens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.generate(new_tokens)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out["mpt_generate"] = {"metric": "clips/sec CLAP embed + MPT-1B prefill + %d-token greedy decode" % new_tokens, "value": round(a.batch / dt, 3), "unit": "clips/s",
                               "ms_per_step": round(dt * 1e3, 2), "clips_per_step": a.batch, "steps": steps, "warmup": 1,
                               "config": {"workload": "configs[4] per-GPU share: 8 clips, CLAP HTSAT-base -> (1,512) -> projector -> MPT-1B (S=132) + %d decode steps" % new_tokens,
                                          "llm_precision": args.llm_precision}}
        del wl
        c = copy.copy(args)
        c.batch = 64
        cw = bench_support.ClapWorkload(c, device, "fp32" if args.llm_precision == "split" else "bf16")
        cw.embed()
        torch.cuda.synchronize()
        ops.start_kernel_timing()
        t0 = time.perf_counter()
        for _ in range(steps):
            cw.embed()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        timers = ops.stop_kernel_timing()
        c.steps = steps
        out["clap_embed"] = {"metric": "clips/sec CLAP HTSAT-base audio embedding (waveform -> 512-d)", "value": round(c.batch / dt, 1), "unit": "clips/s",
                             "ms_per_step": round(dt * 1e3, 3), "clips_per_step": c.batch, "steps": steps, "warmup": 1, "roofline": roofline_clap(timers, c, cw)}
        del cw
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU per step (default 8 = BASELINE configs[1]; 4 for --stages train = configs[3]: 32 clips over 8 GPUs)")
    ap.add_argument("--stages", default="e2e", choices=["e2e", "jukebox", "llama", "train", "generate", "mpt", "mpt-train", "clap", "dist-check"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL; gloo only for --stages dist-check on CPU)")
    ap.add_argument("--new-tokens", type=int, default=64, help="generate: answer tokens per clip (BASELINE configs[2]: 64)")
    ap.add_argument("--micro-batch", type=int, default=4, help="train: clips per micro-step (train_llark.sh uses per_device_train_batch_size 2 x accumulation; 4 x 1 is the same optimizer step, measured 24 %% faster)")
    ap.add_argument("--loss-groups", type=int, default=1, help="train: recipe micro-batches fused into one pass of --micro-batch clips (loss normalised per group: the same optimizer step)")
    ap.add_argument("--grad-checkpoint", action="store_true", help="train: per-layer recompute in the backward (train_llark.sh:25 --gradient_checkpointing True)")
    ap.add_argument("--train-seq", type=int, default=512, help="train: tokens per clip (371 prompt+audio positions + answer)")
    ap.add_argument("--grad-comm", dest="grad_comm", default="bf16", choices=["fp32", "bf16"],
                    help="train: transport dtype of the gradient all-reduce (bf16 = what the reference's DDP sends after model.to(bfloat16): "
                         "13.5 GB per step, SURVEY section 8(e); fp32 = 27 GB); accumulation and AdamW stay fp32 either way")
    ap.add_argument("--llm-layers", type=int, default=0, help="debug: override the number of Llama layers (train stage)")
    ap.add_argument("--depth", type=int, default=0, help="debug: override prior depth (result is then NOT the headline)")
    ap.add_argument("--tiny", action="store_true", help="debug: tiny twin model")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the second pass that times the prior in the other precision")
    ap.add_argument("--cpu-layers", type=int, default=2)
    ap.add_argument("--cpu-warm", type=int, default=1, help="cpu_baseline: untimed warm-up runs per stage (BASELINE.md section 2 asks for 3; bounded to keep the default run within minutes)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="cpu_baseline: timed runs per stage, the median is reported (BASELINE.md section 2: 5)")
    ap.add_argument("--no-extras", action="store_true", help="e2e: skip the configs[2] (generate) and configs[3] (train) legs attached as `extra`")
    ap.add_argument("--prior-precision", default=None, choices=["lo8", "f16x2"],
                    help="how the prior's Conv1D products carry the fp32 activation: f16x2 (library default) = fp16 hi + fp16 lo planes, two "
                         "fp16 MFMA passes, 22 significant bits, 36-layer embedding max-abs-err 5.4e-5; lo8 (opt-in) = fp16 hi plane + E4M3 "
                         "low plane (one fp16 pass + one MX-fp8 MFMA; 15-16 bits, 5.1e-4).  The default run times BOTH: `value` is the "
                         "chosen one, `alt_prior_precision` the other")
    ap.add_argument("--llm-precision", default="split", choices=["split", "bf16"],
                    help="Llama half: fp32-class bf16 hi+lo (default, matches the fp32 reference path) or single-pass bf16")
    args = ap.parse_args()
    if not args.batch:
        args.batch = {"train": 4, "mpt-train": 4, "generate": 1, "clap": 64}.get(args.stages, 8)

    from llark_amd import dist as D

    if D.needs_self_launch(args.gpus):
        # started plainly with --gpus N > 1: become the launcher of N ranks (one per GPU) and relay their exit code
        sys.exit(D.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if args.stages == "dist-check":
        return dist_check(args)
    if args.backend != "nccl":
        raise SystemExit("bench.py: --backend gloo is only valid with --stages dist-check (the hot path runs on GPUs over RCCL)")

    rank, world, local = _dist_setup(args.gpus)
    device = torch.device("cuda", local)
    from llark_amd import ops

    hps, weights, enc, audio, llm = build_workload(args, device)

    def step():
        if args.stages in ("train", "mpt-train"):
            return llm.step()
        if args.stages == "mpt":
            return llm.generate(args.new_tokens)
        if args.stages == "clap":
            return llm.embed()
        emb = enc(audio) if args.stages != "llama" else None
        if args.stages == "generate":
            return llm.generate(emb, args.new_tokens)
        if llm is not None:
            return llm.forward(emb)
        return emb

    with torch.no_grad():
        if args.stages in ("train", "mpt-train"):
            torch.cuda.reset_peak_memory_stats()
        for _ in range(args.warmup):
            step()
        _barrier(world)
        # Per-op HIP events are recorded inside the timed region (they are where `roofline` comes from) -- except for `generate` / `mpt`,
        # whose decode steps are ~160 small launches per token: there the events alone cost ~0.8 ms per token (4.8 vs 4.0 ms,
        # scripts/bench_decode.py), so the timed region runs uninstrumented and the per-kernel breakdown comes from a second,
        # identical pass.
        events_in_timed = args.stages not in ("generate", "mpt")
        if events_in_timed:
            ops.start_kernel_timing()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0               # this rank's own K steps (before waiting for the others)
        _barrier(world)
        elapsed = time.perf_counter() - t0
        if events_in_timed:
            timers = ops.stop_kernel_timing()
        else:
            ops.start_kernel_timing()
            for _ in range(args.steps):
                step()
            timers = ops.stop_kernel_timing()

    elapsed = D.max_over_ranks(elapsed, world, device=device)
    per_rank_ms = D.gather_floats(mine / args.steps * 1e3, world, device=device)
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as tdist

        rccl_ranks = tdist.get_world_size()

    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips / elapsed
        # dominant kernel: the split-fp16 MFMA GEMM of the prior.  Algorithmic flops (2*M*N*K of the
        # fp32-equivalent product; the hi/lo second pass is NOT counted) / HIP-event time of its launches.
        roof = None
        if args.stages != "llama" and enc is not None and prior_roofline(timers, hps, args, args.prior_precision) is not None:
            roof = prior_roofline(timers, hps, args, args.prior_precision)
        elif args.stages == "clap":
            roof = roofline_clap(timers, args, llm)
        elif llm is not None and args.stages not in ("mpt", "mpt-train"):
            roof = llm.roofline(timers, args)
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.stages not in ("mpt", "mpt-train"):
            if args.stages == "clap":
                cpu = cpu_baseline_clap(llm)
            elif args.stages == "train":
                cpu = cpu_baseline_train(args)
            else:
                cpu = cpu_baseline(hps, weights, args)
        workload = {"e2e": "configs[1]+llm: 8x25s clips -> Jukebox VQ-VAE+36-layer prior -> 240x4800 -> projector -> Llama-2-7B fwd (S=371) logits",
                    "jukebox": "configs[1]: Jukebox encoder+prior forward, batch=8x25s clips -> (240,4800) embeddings",
                    "llama": "projector + Llama-2-7B forward (S=371) on precomputed embeddings",
                    "generate": "configs[2]: %d clip(s) -> Jukebox embed -> projector -> Llama-2-7B prefill (S=371) + %d greedy decode steps (KV cache, stopping criterion off)" % (args.batch, args.new_tokens),
                    "mpt": "configs[4]: 10 s 48 kHz clips -> log-mel -> CLAP HTSAT-base -> (1,512) embedding -> projector -> MPT-1B prefill (S=132) + %d greedy decode steps" % args.new_tokens,
                    "clap": "configs[4] audio half: %d x 10 s 48 kHz clips -> int16 round trip + log-mel (1001x64) -> CLAP HTSAT-base -> (B,512) unit embeddings" % args.batch,
                    "mpt-train": "MPT-1B instruction-tuning step on CLAP-style (1,512) embeddings: per GPU %d clips x %d tokens, fwd+bwd+all-reduce+grad-norm clip 1.0+AdamW (train_mpt_model.sh analogue of configs[3])" % (args.batch, args.train_seq),
                    "train": "configs[3]: instruction-tuning step, random-init Llama-2-7B + projector on frozen Jukebox features; per GPU %d clips = %d x %d accumulation micro-steps, S=%d; fwd+bwd+grad all-reduce+grad-norm clip 1.0 (HF Trainer default)+AdamW"
                             % (args.batch, args.micro_batch, max(1, args.batch // args.micro_batch), args.train_seq)}[args.stages]
        line = {
            "metric": {"train": "clips/sec instruction-tuning step (fwd+bwd+all-reduce+AdamW)",
                       "generate": "clips/sec embed + prefill + %d-token greedy decode" % args.new_tokens,
                       "mpt": "clips/sec CLAP embed + MPT-1B prefill + %d-token greedy decode" % args.new_tokens,
                       "clap": "clips/sec CLAP HTSAT-base audio embedding (waveform -> 512-d)",
                       "mpt-train": "clips/sec MPT-1B instruction-tuning step (fwd+bwd+all-reduce+AdamW)"}.get(args.stages, "clips/sec (25 s audio + 128-tok prompt) end-to-end fwd"),
            "value": round(value, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "rccl_ranks": rccl_ranks, "per_rank_ms_per_step": [round(v, 3) for v in per_rank_ms],
            "vs_baseline": None, "dtype": {"e2e": PRIOR_DT[args.prior_precision or "f16x2"] + " prior + " + ("bf16x2-split(fp32-class)" if args.llm_precision == "split" else "bf16") + " llm",
                                           "jukebox": PRIOR_DT[args.prior_precision or "f16x2"],
                                           "llama": "bf16x2-split(fp32-class)" if args.llm_precision == "split" else "bf16",
                                           "generate": PRIOR_DT[args.prior_precision or "f16x2"] + " prior + " + ("bf16x2-split(fp32-class)" if args.llm_precision == "split" else "bf16") + " llm",
                                           "mpt": "bf16x2-split(fp32-class)" if args.llm_precision == "split" else "bf16",
                                           "clap": "bf16x2-split activations x bf16x2-split weights (fp32-class)" if args.llm_precision == "split" else "bf16",
                                           "mpt-train": "bf16 (fp32 accumulate, fp32 grads + AdamW moments)",
                                           "train": "bf16 (fp32 accumulate, fp32 grads + AdamW moments)"}[args.stages],
            "data": "synthetic",
            "config": {"workload": workload, "clips_per_gpu": args.batch, "global_batch": args.batch * world,
                       "audio_samples": 480000 if args.stages in ("clap", "mpt") else hps.sample_length, "prompt_tokens": 0 if args.stages == "clap" else 128, "seq_len": {"train": args.train_seq, "mpt-train": args.train_seq, "mpt": 132}.get(args.stages, 371),
                       "prior_depth": None if args.stages in ("clap", "mpt", "mpt-train") else hps.prior_depth,
                       "prior_precision": getattr(args, "prior_precision", None) if args.stages in ("e2e", "jukebox", "generate") else None,
                       "prior_layernorm_folded": getattr(args, "prior_ln_fold", None) if args.stages in ("e2e", "jukebox", "generate") else None,
                       "parallelism": f"dp{world} (clip-sharded, no collective)" if args.stages != "train" else f"dp{world} (clip-sharded, RCCL all-reduce of fp32 gradients ({args.grad_comm} on the links) once per optimizer step)",
                       "debug_overrides": bool(args.depth or args.tiny)},
            "roofline": roof, "roofline_conv": conv_roofline(timers, args),
            "roofline_llm": (llm.roofline(timers, args) if llm is not None and args.stages not in ("train", "mpt", "mpt-train", "clap") else None),
            "cpu_baseline": cpu,
            "kernel_ms": {k: round(v[1] / args.steps, 3) for k, v in timers.items()},
        }
        if args.stages == "llama" and line.get("roofline_llm") and line["roofline_llm"].get("whole_forward_flops_t"):
            line["roofline_llm"]["whole_forward_frac"] = round(line["roofline_llm"]["whole_forward_flops_t"] / (elapsed / args.steps) / PEAK_F16_MFMA_TFLOPS, 4)
        if args.stages in ("generate", "mpt"):
            line["kernel_ms_from"] = "second pass with per-op events (the timed region runs without them)"
        if args.stages in ("e2e", "llama") and llm is not None and world == 1:
            # the Llama half in the OTHER activation precision (same weights), outside the timed region: both MFMA fractions in one line
            other = "bf16" if args.llm_precision == "split" else "split"
            with torch.no_grad():
                ref_logits = llm.forward(None).clone()              # this precision's logits on the bench prompts (N(0,1) audio stand-in)
                llm.engine.set_precision(other)
                oth_logits = llm.forward(None)
                torch.cuda.synchronize()
                # live parity pair: the two flows on the SAME weights and inputs.  "split" is pinned to the fp32
                # oracle at 6.2e-5 of max|logits| (tests/test_fulldepth_gpu.py), so this difference IS the bf16 flow's error to that accuracy.
                lmax = float(ref_logits.abs().max())
                diff = (oth_logits - ref_logits).abs()
                parity_pair = {"against": args.llm_precision, "logits_max_abs_diff": round(float(diff.max()), 6), "logits_max_abs": round(lmax, 4),
                               "diff_over_max": round(float(diff.max()) / lmax, 6),
                               "rms_diff_over_max": round(float(diff.pow(2).mean().sqrt()) / lmax, 7),
                               "argmax_agree_frac": round(float((oth_logits.argmax(-1) == ref_logits.argmax(-1)).float().mean()), 5),
                               "positions": int(ref_logits.shape[0] * ref_logits.shape[1]),
                               "fulldepth_fixture": FULLDEPTH_LLM_PARITY.get(other)}
                del ref_logits, oth_logits, diff
                ops.start_kernel_timing()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    llm.forward(None)
                torch.cuda.synchronize()
                ms_other = (time.perf_counter() - t1) / args.steps * 1e3
                roof_other = llm.roofline(ops.stop_kernel_timing(), args)
                llm.engine.set_precision(args.llm_precision)
            if roof_other is not None:
                roof_other.update({"llm_precision": other, "llama_ms_per_step": round(ms_other, 3), "parity": parity_pair})
                wf = roof_other.get("whole_forward_flops_t")
                if wf and ms_other > 0:             # north_star's ">= 40 % of the MFMA roofline on the Llama forward": all flops / forward time
                    roof_other["whole_forward_frac"] = round(wf / (ms_other * 1e-3) / PEAK_F16_MFMA_TFLOPS, 4)
            line["roofline_llm_" + other] = roof_other
        if args.stages in ("e2e", "jukebox") and enc is not None and world == 1 and not args.no_alt_precision:
            # The SAME step with the prior in the OTHER precision (same weights, second encoder object), outside the timed region, so
            # that one driver-recorded line carries both: `value` belongs to the default f16x2 (22-bit activations, full-depth
            # embedding max-abs-err 5.4e-5 <= 1e-4 = the configs[1] bar); lo8 (15-16 bits, 5.1e-4) is the opt-in speed mode.
            other = "lo8" if args.prior_precision == "f16x2" else "f16x2"
            from llark_amd.jukebox import extract as E

            with torch.no_grad():
                enc_main = enc
                enc2 = E.WrappedAudioEncoder(hps=hps, weights=weights, device=device, precision=other)
                enc2.vqvae.set_codebook(enc_main.vqvae.k)
                enc = enc2
                for _ in range(max(1, args.warmup)):
                    step()
                torch.cuda.synchronize()
                ops.start_kernel_timing()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                dt_other = time.perf_counter() - t1
                timers_other = ops.stop_kernel_timing()
                enc = enc_main
                del enc2
            line["alt_prior_precision"] = {
                "prior_precision": other, "dtype": PRIOR_DT[other], "value": round(args.batch * args.steps / dt_other, 4), "unit": "clips/s",
                "ms_per_step": round(dt_other / args.steps * 1e3, 3), "roofline": prior_roofline(timers_other, hps, args, other),
                "note": ("opt-in mode, narrower than the reference's fp32 activations: 36-layer B=8 embedding max-abs-err 5.1e-4 (> the 1e-4 of "
                         "configs[1]; tests/test_fulldepth_gpu.py) -- reported next to the headline, never as it") if other == "lo8" else
                        "the library default (22-bit activations; full-depth embedding max-abs-err 5.4e-5 <= 1e-4)"}
        if args.stages in ("e2e", "jukebox") and enc is not None and world == 1:
            # parity of the benchmarked encoder on the bench's own clips, outside the timed region: the default path (fused stages +
            # near-tie certificate, whose fix-up ran INSIDE the timed region above) against the exact per-layer kernels
            from llark_amd.jukebox.vqvae import VQVAE

            with torch.no_grad():
                vq = enc.vqvae
                got = vq.encode_top(audio)
                ties = vq.last_near_ties
                ex = VQVAE(hps, weights, device, exact=True)
                ex.set_codebook(vq.k)
                exact_codes = ex.encode_top(audio)
                # the same fused stages WITHOUT the certificate (round 3's default), for the record: what exactness costs
                raw = VQVAE(hps, weights, device, tie_e_rel=None)
                raw.set_codebook(vq.k)
                raw_codes = raw.encode_top(audio)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    raw.encode_top(audio)
                e1.record()
                torch.cuda.synchronize()
                raw_ms = e0.elapsed_time(e1) / 5
                raw_gbs = raw.algorithmic_bytes(audio.shape[0]) / (raw_ms * 1e-3) / 1e9
                del raw
                line["vq_codes"] = {"code_mismatches_vs_exact": int((got != exact_codes).sum()), "tokens": int(got.numel()),
                                    "without_certificate": {"code_mismatches_vs_exact": int((raw_codes != exact_codes).sum()), "ms": round(raw_ms, 3),
                                                            "frac_of_hbm_roofline": round(raw_gbs / PEAK_HBM_GBS, 4),
                                                            "note": "VQVAE(tie_e_rel=None): the fused stages + plain argmin, timed outside the timed region"},
                                    "near_tie_tokens_reevaluated_exactly": int(ties), "tie_e_rel": vq.tie_e_rel, "tie_ulps": vq.tie_ulps,
                                    "note": "default encoder = fused split-fp16 stages + near-tie certificate; flagged tokens are re-evaluated by the exact "
                                            "kernels on receptive-field windows inside the timed region (roofline_conv includes it)"}
                del ex
        if args.stages == "e2e" and world == 1 and not args.no_extras and not (args.tiny or args.depth or args.llm_layers):
            line["extra"] = {}
            try:
                line["extra"]["generate"] = extra_generate(args, enc, audio, llm, cpu)
            except Exception as e:  # noqa: BLE001 -- an extra leg must not take the headline line down with it
                line["extra"]["generate"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                del enc, llm
                step = None
                torch.cuda.empty_cache()
                line["extra"].update(extra_mpt(args, device))
            except Exception as e:  # noqa: BLE001
                line["extra"]["mpt_generate"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                enc = llm = None
                torch.cuda.empty_cache()
                line["extra"]["train"] = extra_train(args, device)
            except Exception as e:  # noqa: BLE001
                line["extra"]["train"] = {"error": f"{type(e).__name__}: {e}"}
            llm = None
        if args.stages in ("train", "mpt-train"):
            line["peak_hbm_gb"] = round(torch.cuda.max_memory_allocated() / 2**30, 1)
        if args.stages == "train":
            # MFU = 6 x parameters x tokens / step time / dense bf16 MFMA peak (per GPU); and the part of the gradient exchange the
            # backward did not hide (HIP events around allreduce_grads on the compute stream; 0 on one GPU) next to the step time
            line["mfu"] = round(llm.model_flops_per_step() / (elapsed / args.steps) / (PEAK_F16_MFMA_TFLOPS * 1e12), 4)
            line["allreduce_exposed_ms"] = round(llm.trainer.exposed_exchange_ms() / max(1, args.steps + args.warmup), 3)
        if args.stages == "generate":
            key = "gemm_split_bf16_skinny" if args.llm_precision == "split" else "gemm_bf16_skinny"
            if key in timers:
                launches, ms, _ = timers[key]
                nsteps = (args.new_tokens - 1) * args.steps
                gbs = llm.decode_weight_bytes() * nsteps / (ms * 1e-3) / 1e9
                line["roofline_decode"] = {"bound": "hbm", "kernel": "gemm_skinny_kernel (M <= 16 weight streaming)", "achieved": round(gbs, 1),
                                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                                           "launches": launches, "avg_launch_ms": round(ms / launches, 4),
                                           "decode_ms_per_token": round(sum(v[1] for k, v in timers.items() if k.endswith("_skinny")) / nsteps, 3)}
        print(json.dumps(line), flush=True)
    D.shutdown(world)


if __name__ == "__main__":
    main()
