"""GPU parity: the training step (forward with saved activations, full backward, AdamW) vs torch autograd on the
oracle evaluated in the same bf16 flow (reference recipe: bf16, lm_head frozen, embeddings trainable only on the
<audio_start>/<audio_end> rows, projector + every Llama weight trainable).

Tolerance: gradients are compared per tensor by relative Frobenius error <= 3e-2 and cosine >= 0.999 (bf16 operands
in both the forward and the backward products; the oracle back-propagates in fp32 through bf16-rounded activations)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.bfloat16().float()


def _setup(B=2, layers=2):
    from llark_amd.m2t.engine import HipLlamaEngine, LlamaDims
    from oracle import llama_ref as LR
    V = 128
    spec = LR.LlamaSpec(hidden_size=256, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=2, vocab_size=V,
                        mm_hidden_size=96, audio_start_token=V - 2, audio_end_token=V - 1, audio_patch_token=V - 3)
    w = {k: _bf(v) for k, v in LR.make_weights(spec, seed=0, std=0.08).items()}
    g = torch.Generator().manual_seed(5)
    F = 5
    ids = torch.stack([torch.tensor([1] + torch.randint(3, V - 3, (2 + b,), generator=g).tolist() + [V - 2] + [V - 3] * F + [V - 1]
                                    + torch.randint(3, V - 3, (9 - b,), generator=g).tolist()) for b in range(B)])
    aud = torch.randn(B, F, 96, generator=g)
    labels = ids.clone()
    labels[:, :10] = -100
    dims = LlamaDims(hidden_size=256, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=2, vocab_size=V,
                     mm_hidden_size=96)
    eng = HipLlamaEngine(dims, "cuda", B, 64, precision="bf16")
    eng.load_state_dict(w)
    segs = [(b, int((ids[b] == V - 2).nonzero()[0, 0]), aud[b].cuda()) for b in range(B)]
    return spec, w, ids, aud, labels, eng, segs


def _oracle_grads(spec, w, ids, aud, labels):
    from oracle import llama_ref as LR
    wp = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    out = LR.forward(wp, spec, ids, aud, labels=labels, act_dtype=torch.bfloat16, round_probs=True)
    out["loss"].backward()
    return out["loss"].item(), {k: v.grad for k, v in wp.items()}


def test_forward_backward_grads_match_autograd():
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup()
    tr = HipLlamaTrainer(eng, embed_grad_tokens=(spec.audio_start_token, spec.audio_end_token))
    loss = tr.forward_backward(ids.cuda(), segs, labels.cuda()).item()
    ref_loss, ref = _oracle_grads(spec, w, ids, aud, labels)
    assert abs(loss - ref_loss) <= 5e-3 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    got = tr.export_grads_hf()
    worst = {}
    for name, gh in got.items():
        r = ref[name]
        gh = gh.float().cpu()
        if name == "model.embed_tokens.weight":
            # orig_embeds_params semantics: only the <audio_start>/<audio_end> rows receive gradient
            rows = [spec.audio_start_token, spec.audio_end_token]
            others = [i for i in range(spec.vocab_size) if i not in rows]
            assert gh[others].abs().max().item() == 0.0
            gh, r = gh[rows], r[rows]
        rel = ((gh - r).norm() / (r.norm() + 1e-30)).item()
        cos = torch.nn.functional.cosine_similarity(gh.flatten(), r.flatten(), dim=0).item()
        worst[name] = (rel, cos)
        assert np.isfinite(rel) and rel <= 3e-2 and cos >= 0.999, f"{name}: rel {rel:.3e} cos {cos:.5f}"
    top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:3]
    print("worst grad rel errs:", [(k, f"{v[0]:.2e}") for k, v in top])
    assert "lm_head.weight" not in got                                   # frozen like the reference


@pytest.mark.parametrize("dx_direct_uses", [None, 1])
def test_accumulation_and_adamw_step(dx_direct_uses):
    """Two micro-batches with loss_scale 1/2 == one batch of both; then one AdamW step == torch.optim.AdamW.
    dx_direct_uses=1: the second micro-batch takes its dX products through the W^T / fragment-major twins (default: W as stored)."""
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup(B=2)
    tr = HipLlamaTrainer(eng, lr=1e-2, weight_decay=0.0, embed_grad_tokens=(spec.audio_start_token, spec.audio_end_token))
    if dx_direct_uses is not None:
        tr.dx_direct_uses = dx_direct_uses
    tr.forward_backward(ids.cuda(), segs, labels.cuda())
    g_full = tr.flat_grad.clone()
    tr.zero_grad()
    # same batch twice with scale 1/2 must reproduce the gradient (accumulation is a plain sum)
    tr.forward_backward(ids.cuda(), segs, labels.cuda(), loss_scale=0.5)
    tr.forward_backward(ids.cuda(), segs, labels.cuda(), loss_scale=0.5)
    rel = ((tr.flat_grad - g_full).norm() / g_full.norm()).item()
    assert rel < 2e-2, rel
    # AdamW: first step moves every parameter with a non-zero gradient by ~lr * sign(g)
    name = "layers.0.wo"
    p_before = dict(tr.params)[name].float().clone()
    gsel = tr.grads[name].clone()
    tr.step()
    p_after = dict(tr.params)[name].float()
    ref_p = torch.nn.Parameter(p_before.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    ref_p.grad = gsel.clone()
    opt.step()
    assert (p_after - ref_p.detach().bfloat16().float()).abs().max().item() <= 2 ** -7 * p_before.abs().max().item() + 1e-3
    assert tr.step_count == 1
    # after the step the gradients start over: accumulate-only slices are zero, matrix gradients are OVERWRITTEN by the
    # next micro-batch (no memset), and a matrix whose dW never runs (text-only batch: mm_projector) reads as zero
    assert tr.grads["norm"].abs().max().item() == 0.0 and tr.grads["embed"].abs().max().item() == 0.0
    stale = tr.grads["proj_w"].clone()
    assert stale.abs().max().item() > 0.0                                   # previous step's values are still in the buffer
    tr.forward_backward(ids.cuda(), [], labels.cuda())                      # no audio segments: projector untouched
    g2 = tr.export_grads_hf()
    assert g2["model.mm_projector.weight"].abs().max().item() == 0.0
    tr2 = HipLlamaTrainer(eng, lr=1e-2, weight_decay=0.0, embed_grad_tokens=(spec.audio_start_token, spec.audio_end_token))
    tr2.forward_backward(ids.cuda(), [], labels.cuda())
    for name, prm in tr.params:                                            # identical to a freshly zeroed trainer
        if prm.dim() == 2:
            assert torch.equal(tr.grads[name], tr2.grads[name]), name
        else:                                                              # gain gradients are summed with fp32 atomics (order varies)
            assert torch.allclose(tr.grads[name], tr2.grads[name], rtol=1e-5, atol=1e-7), name


def test_gradient_checkpointing_gives_identical_gradients_with_less_memory():
    """train_llark.sh:25 `--gradient_checkpointing True`: only each layer's input is kept, the backward re-runs that layer's
    forward with the same kernels in the same order -> the same gradients (matrix gradients bit for bit; the RMSNorm gain
    gradients are fp32 atomic sums), the same loss, and a lower activation peak."""
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup(B=2)
    toks = (spec.audio_start_token, spec.audio_end_token)
    peaks, grads, losses = [], [], []
    for ckpt in (False, True):
        tr = HipLlamaTrainer(eng, lr=1e-2, embed_grad_tokens=toks, gradient_checkpointing=ckpt, optimizer_state=False)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        losses.append(float(tr.forward_backward(ids.cuda(), segs, labels.cuda())))
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated() - base)
        grads.append({n: tr.grads[n].clone() for n, _ in tr.params})
        del tr
    assert losses[0] == losses[1]
    for (name, prm) in HipLlamaTrainer(eng, embed_grad_tokens=toks, optimizer_state=False).params:
        a, b = grads[0][name], grads[1][name]
        if prm.dim() == 2 and name != "embed":
            assert torch.equal(a, b), name
        else:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), name
    assert peaks[1] < peaks[0], peaks
    with pytest.raises(RuntimeError, match="optimizer_state=False"):
        HipLlamaTrainer(eng, embed_grad_tokens=toks, optimizer_state=False).step()


def test_training_reduces_loss():
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup(B=2)
    tr = HipLlamaTrainer(eng, lr=2e-3, embed_grad_tokens=(spec.audio_start_token, spec.audio_end_token))
    losses = []
    for _ in range(6):
        losses.append(tr.forward_backward(ids.cuda(), segs, labels.cuda()).item())
        tr.step()
    assert losses[-1] < 0.7 * losses[0], losses
    # the engine used for inference sees the updated weights (same tensors)
    lg = eng.forward_tokens(ids.cuda(), segs)
    assert torch.isfinite(lg).all()


def test_wrapped_model_loss_backward_dropin():
    """The reference's call pattern: loss = model(input_ids, labels, audio_encodings).loss; loss.backward();
    torch optimizer step -- on the HIP training step through the autograd bridge."""
    from llark_amd.m2t.llamav2 import WrappedLlamav2Config, WrappedLlamav2ForCausalLM
    spec, w, ids, aud, labels, eng, segs = _setup(B=2)
    del eng
    cfg = WrappedLlamav2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                               num_key_value_heads=2, vocab_size=spec.vocab_size, max_position_embeddings=512,
                               rms_norm_eps=1e-5, tie_word_embeddings=False)
    cfg.mm_hidden_size = 96
    m = WrappedLlamav2ForCausalLM(cfg)
    m.get_model().initialize_adapter_modules()
    m.load_state_dict(w, strict=False)
    ac = m.get_model().audio_encoder_config
    ac.audio_start_token, ac.audio_end_token, ac.audio_patch_token = spec.audio_start_token, spec.audio_end_token, spec.audio_patch_token
    m.get_model().orig_embeds_params = [m.get_input_embeddings().weight.data.clone()]   # tune_mm_mlp_adapter semantics
    for p in m.get_output_embeddings().parameters():
        p.requires_grad = False                                                           # lm_head frozen (llamav2.py:414)
    m.cuda().train()
    m.configure_engine(max_batch=2, max_seq=64)
    out = m(input_ids=ids.cuda(), labels=labels.cuda(), audio_encodings=aud.cuda())
    out.loss.backward()
    ref_loss, ref = _oracle_grads(spec, w, ids, aud, labels)
    assert abs(out.loss.item() - ref_loss) <= 5e-3 * max(1.0, abs(ref_loss))
    for name in ("model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight", "model.mm_projector.weight",
                 "model.mm_projector.bias", "model.norm.weight"):
        gp = dict(m.named_parameters())[name].grad.float().cpu()
        rel = ((gp - ref[name]).norm() / ref[name].norm()).item()
        assert rel <= 3e-2, f"{name}: {rel:.3e}"
    assert m.lm_head.weight.grad is None
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=2e-3)
    l0 = out.loss.item()
    for _ in range(5):
        opt.step()
        opt.zero_grad()
        out = m(input_ids=ids.cuda(), labels=labels.cuda(), audio_encodings=aud.cuda())
        out.loss.backward()
    assert out.loss.item() < 0.8 * l0, (l0, out.loss.item())


def test_train_loop_schedule():
    from llark_amd.m2t import AudioEncoderConfig
    from llark_amd.m2t.train import TrainConfig, lr_at, train
    spec, w, ids, aud, labels, eng, segs = _setup(B=2)
    cfg = TrainConfig(learning_rate=2e-3, max_steps=20, gradient_accumulation_steps=2)
    assert lr_at(0, cfg) == 0.0 and abs(lr_at(1, cfg) - 2e-3) < 1e-12 and lr_at(20, cfg) == 0.0     # warm-up = ceil(0.6) = 1 step
    ac = AudioEncoderConfig()
    ac.audio_start_token, ac.audio_end_token, ac.audio_patch_token = spec.audio_start_token, spec.audio_end_token, spec.audio_patch_token
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids, dtype=torch.bool), audio_encodings=aud)
    logs = []
    losses = train(eng, [batch] * 12, ac, cfg, world=1, log=logs.append)
    assert len(losses) == 6 and logs[-1]["step"] == 6
    assert losses[-1] < losses[1]


def test_train_loop_fused_accumulation_follows_the_separate_micro_batches():
    """TrainConfig.fuse_accumulation: the loop runs a step's micro-batches as one pass (engine sized for them) -- the logged losses track
    the unfused loop's to the fp32-order differences a few optimizer steps amplify; an engine too small for the fused batch falls back to
    separate passes (identical losses)."""
    from llark_amd.m2t import AudioEncoderConfig
    from llark_amd.m2t.engine import HipLlamaEngine
    from llark_amd.m2t.train import TrainConfig, train
    spec, w, ids, aud, labels, eng, segs = _setup(B=2)
    ac = AudioEncoderConfig()
    ac.audio_start_token, ac.audio_end_token, ac.audio_patch_token = spec.audio_start_token, spec.audio_end_token, spec.audio_patch_token
    batch = dict(input_ids=ids, labels=labels, attention_mask=torch.ones_like(ids, dtype=torch.bool), audio_encodings=aud)
    batch2 = dict(batch, input_ids=ids.flip(0), labels=labels.flip(0), audio_encodings=aud.flip(0))
    stream = [batch, batch2] * 4

    def run(fuse, max_batch):
        e = HipLlamaEngine(eng.dims, "cuda", max_batch, eng.smax, precision="bf16")
        e.load_state_dict(w)
        return train(e, stream, ac, TrainConfig(learning_rate=2e-3, max_steps=20, gradient_accumulation_steps=2, fuse_accumulation=fuse), world=1)

    base, fused, small = run(False, 2), run(True, 4), run(True, 2)
    assert len(base) == len(fused) == 4 and small == base
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(base, fused)), (base, fused)
    assert fused[-1] < fused[1]


def test_checkpoint_resume_and_adapter_sidefile(tmp_path):
    """Save after 2 optimizer steps, resume in a NEW engine + trainer, take a 3rd step == 3 uninterrupted steps; the
    on-disk names are the reference's (full model + mm_projector side-file), old checkpoints are pruned."""
    from llark_amd.m2t import checkpoint as CK
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup(B=2)
    toks = (spec.audio_start_token, spec.audio_end_token)
    tr = HipLlamaTrainer(eng, lr=2e-3, embed_grad_tokens=toks)
    out = str(tmp_path / "run")
    for step in range(3):
        tr.forward_backward(ids.cuda(), segs, labels.cuda())
        tr.step()
        if step < 2:
            CK.save_checkpoint(tr, out, save_total_limit=1)
    final = {k: v.float().cpu().clone() for k, v in eng.state_dict_hf().items()}
    assert [os.path.basename(p) for p in CK.list_checkpoints(out)] == ["checkpoint-2"]          # checkpoint-1 pruned
    side = torch.load(os.path.join(out, "mm_projector", "checkpoint-2.bin"))
    assert sorted(side) == ["model.embed_tokens.weight", "model.mm_projector.bias", "model.mm_projector.weight"]
    full = torch.load(os.path.join(out, "checkpoint-2", "pytorch_model.bin"))
    assert set(full) == set(w) and all(full[k].shape == w[k].shape for k in w)                  # the reference's names / shapes
    # resume: fresh engine with the ORIGINAL weights, then load
    spec2, w2, ids2, aud2, labels2, eng2, segs2 = _setup(B=2)
    tr2 = HipLlamaTrainer(eng2, lr=2e-3, embed_grad_tokens=toks)
    assert CK.maybe_resume(tr2, out) == 2 and tr2.step_count == 2
    tr2.forward_backward(ids.cuda(), segs2, labels.cuda())
    tr2.step()
    got = {k: v.float().cpu() for k, v in eng2.state_dict_hf().items()}
    for k in final:
        assert torch.allclose(got[k], final[k], rtol=0, atol=2 ** -8 * max(1e-3, final[k].abs().max().item())), k
    # the wrapped model of the inference path loads the saved file under the same names
    assert CK.maybe_resume(HipLlamaTrainer(_setup(B=2)[5], lr=2e-3, embed_grad_tokens=toks), str(tmp_path / "empty")) == 0


def test_native_loop_from_tar_shards(tmp_path):
    """m2t/train.py path end to end on synthetic shards in the reference's format: tar members (<key>.json +
    <key>.audio_encoding.npy) -> conversations -> prompt glue -> collated micro-batches -> HIP training steps with gradient
    accumulation -> checkpoint; the loss falls and a re-run resumes from the saved step."""
    import io
    import json
    import sys
    import tarfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from toy_tokenizer import ToyTokenizer
    from llark_amd.m2t import AudioEncoderConfig
    from llark_amd.m2t import checkpoint as CK
    from llark_amd.m2t.data import micro_batches
    from llark_amd.m2t.engine import HipLlamaEngine, LlamaDims
    from llark_amd.m2t.prompting import DEFAULT_CONVERSATION_HEADER
    from llark_amd.m2t.train import TrainConfig, train
    from oracle import llama_ref as LR
    rng = np.random.default_rng(0)
    words = ["slow", "fast", "jazz", "rock", "piano", "drums"]
    for s in range(2):
        with tarfile.open(tmp_path / f"train-{s:03d}.tar", "w") as tf:
            for k in range(3):
                resp = {"response": [{"question": "what is the genre ?", "answer": f"it is {words[(s + k) % 6]} ."}]}
                for name, payload in ((f"c{s}{k}.json", json.dumps(resp).encode()),):
                    info = tarfile.TarInfo(name)
                    info.size = len(payload)
                    tf.addfile(info, io.BytesIO(payload))
                b = io.BytesIO()
                np.save(b, rng.standard_normal((4, 96)).astype(np.float32))
                info = tarfile.TarInfo(f"c{s}{k}.audio_encoding.npy")
                info.size = len(b.getvalue())
                tf.addfile(info, io.BytesIO(b.getvalue()))
    tok = ToyTokenizer()
    for text in (DEFAULT_CONVERSATION_HEADER, "### Human: Assistant: what is the genre ? it is . \n " + " ".join(words)):
        tok.encode(text)
    tok.add_tokens(["<audio_patch>", "<audio_start>", "<audio_end>"], special_tokens=True)
    V = len(tok)
    spec = LR.LlamaSpec(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, vocab_size=V, mm_hidden_size=96)
    w = {k: _bf(v) for k, v in LR.make_weights(spec, seed=2, std=0.08).items()}
    dims = LlamaDims(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, vocab_size=V, mm_hidden_size=96)
    ac = AudioEncoderConfig()
    ac.use_audio_start_end = True
    ac.audio_patch_token, ac.audio_start_token, ac.audio_end_token = tok.convert_tokens_to_ids(["<audio_patch>", "<audio_start>", "<audio_end>"])
    mm = dict(is_multimodal=True, sep_audio_conv_front=False, use_audio_start_end=True)
    out = str(tmp_path / "run")

    def run(max_steps):
        eng = HipLlamaEngine(dims, "cuda", 2, 128, precision="bf16", frag_weights=False)
        eng.load_state_dict(w)
        batches = micro_batches(str(tmp_path / "train-{000..001}.tar"), tok, mm, batch_size=2, model_max_length=128, seed=1)
        cfg = TrainConfig(learning_rate=3e-3, max_steps=40, gradient_accumulation_steps=2)
        logs = []
        train(eng, batches, ac, cfg, world=1, max_optimizer_steps=max_steps, log=logs.append, output_dir=out, save_steps=4, save_total_limit=2)
        return logs

    logs = run(8)
    assert [r["step"] for r in logs] == list(range(1, 9))
    assert logs[-1]["loss"] < 0.8 * logs[0]["loss"], [round(r["loss"], 3) for r in logs]
    assert [os.path.basename(p) for p in CK.list_checkpoints(out)] == ["checkpoint-4", "checkpoint-8"]
    logs2 = run(10)                                                       # a fresh process state resumes at step 8
    assert [r["step"] for r in logs2] == [9, 10]


def test_train_cli_end_to_end(tmp_path):
    """`python -m llark_amd.m2t.train` (the m2t/train.py entry point, reference flag names) on a tiny HF-format checkpoint +
    tokenizer saved locally and synthetic shards: parses the flags, loads model + tokenizer, sets up the audio tokens, runs
    optimizer steps on the HIP training step, writes checkpoint-N/ + mm_projector/checkpoint-N.bin, and a second invocation
    resumes."""
    import io
    import json
    import tarfile
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from llark_amd.m2t import checkpoint as CK
    from llark_amd.m2t import train as T
    from llark_amd.m2t.llamav2 import WrappedLlamav2Config, WrappedLlamav2ForCausalLM
    from llark_amd.m2t.prompting import DEFAULT_CONVERSATION_HEADER
    # ---- tiny tokenizer + checkpoint in HF format
    words = sorted(set((DEFAULT_CONVERSATION_HEADER + " ### Human: Assistant: what is the genre ? it is jazz rock piano . <audio>").split()))
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for wd in words:
        vocab.setdefault(wd, len(vocab))
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    ckpt = tmp_path / "ckpt"
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>")
    fast.save_pretrained(str(ckpt))
    cfg = WrappedLlamav2Config(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                               vocab_size=len(vocab), max_position_embeddings=256, rms_norm_eps=1e-5, tie_word_embeddings=False)
    torch.manual_seed(0)
    WrappedLlamav2ForCausalLM(cfg).save_pretrained(str(ckpt))
    # ---- shards
    rng = np.random.default_rng(0)
    with tarfile.open(tmp_path / "s-000.tar", "w") as tf:
        for k in range(4):
            payload = json.dumps({"response": [{"question": "what is the genre ?", "answer": "it is jazz ."}]}).encode()
            info = tarfile.TarInfo(f"k{k}.json")
            info.size = len(payload)
            tf.addfile(info, io.BytesIO(payload))
            b = io.BytesIO()
            np.save(b, rng.standard_normal((3, 96)).astype(np.float32))
            info = tarfile.TarInfo(f"k{k}.audio_encoding.npy")
            info.size = len(b.getvalue())
            tf.addfile(info, io.BytesIO(b.getvalue()))
    out = tmp_path / "out"
    argv = ["--model_name_or_path", str(ckpt), "--train_data_path", str(tmp_path / "s-{000..000}.tar"), "--output_dir", str(out),
            "--mm_hidden_size", "96", "--mm_use_audio_start_end", "True", "--per_device_train_batch_size", "2",
            "--gradient_accumulation_steps", "2", "--learning_rate", "1e-3", "--max_steps", "3", "--model_max_length", "128",
            "--save_steps", "2", "--save_total_limit", "2", "--bf16", "True", "--report_to", "none"]
    T.main(argv)
    assert [os.path.basename(p) for p in CK.list_checkpoints(str(out))] == ["checkpoint-2", "checkpoint-3"]
    side = torch.load(out / "mm_projector" / "checkpoint-3.bin")
    assert "model.mm_projector.weight" in side and side["model.mm_projector.weight"].shape == (256, 96)
    full = torch.load(out / "checkpoint-3" / "pytorch_model.bin")
    assert full["model.embed_tokens.weight"].shape[0] == len(vocab) + 1 + 3          # + [PAD] + the three audio tokens
    T.main([("4" if a == "3" else a) for a in argv])                     # --max_steps 4: resumes from checkpoint-3, one more step
    assert os.path.basename(CK.latest_checkpoint(str(out))) == "checkpoint-4"


def test_grads_at_7b_width_vs_autograd_fixture():
    """parity of the training step AT THE WIDTH IT IS BENCHMARKED -- hidden 4096, 32 heads x 128, intermediate
    11008, vocab 32004, S = 1024, two decoder layers -- against torch autograd over the fp32 CPU oracle in the bf16 flow
    (tests/golden/train7b_grads.npz <- tests/golden/make_train7b_golden.py; weights regenerated from the same seed on both sides).
    Per gradient tensor the fixture holds the Frobenius norm, all row sums, all column sums and a fixed 64 x 64 sample of entries
    (1-D gradients and the two trainable embedding rows whole); each must match like the small-width test's whole tensors:
    relative error <= 3e-2, cosine >= 0.999."""
    import importlib.util

    from llark_amd.m2t.engine import HipLlamaEngine, LlamaDims
    from llark_amd.m2t.train_engine import HipLlamaTrainer

    here = os.path.dirname(os.path.abspath(__file__))
    sp = importlib.util.spec_from_file_location("make_train7b_golden", os.path.join(here, "golden", "make_train7b_golden.py"))
    G = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(G)
    z = np.load(G.NPZ)
    spec, w, ids, labels, aud = G.train7b_setup()
    assert int(z["seq"]) == ids.shape[1] and int(z["layers"]) == spec.num_hidden_layers
    dims = LlamaDims(num_hidden_layers=spec.num_hidden_layers, vocab_size=spec.vocab_size)
    assert (dims.hidden_size, dims.intermediate_size, dims.num_attention_heads) == (4096, 11008, 32)
    eng = HipLlamaEngine(dims, "cuda", 1, ids.shape[1], precision="bf16")
    eng.load_state_dict(w)
    del w
    tr = HipLlamaTrainer(eng, embed_grad_tokens=(spec.audio_start_token, spec.audio_end_token))
    segs = [(0, int((ids[0] == spec.audio_start_token).nonzero()[0, 0]), aud[0].cuda())]
    loss = tr.forward_backward(ids.cuda(), segs, labels.cuda(), last_micro_batch=True).item()
    # S = 1024 rows: the dW products run llark_gemm16_t_sumsq and leave their share of the squared gradient norm behind
    assert len(tr._norm_spans) >= 4 * spec.num_hidden_layers
    tr._finalize_grads()
    full_norm = tr.flat_grad.double().norm().item()
    assert abs(tr.grad_norm() - full_norm) <= 1e-6 * full_norm
    ref_loss = float(z["loss"])
    assert abs(loss - ref_loss) <= 5e-3 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    got = tr.export_grads_hf()

    def close(what, a, b):
        a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
        rel = ((a - b).norm() / (b.norm() + 1e-30)).item()
        cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
        assert np.isfinite(rel) and rel <= 3e-2 and cos >= 0.999, f"{what}: rel {rel:.3e} cos {cos:.5f}"
        return rel

    worst, checked = {}, 0
    for name, gh in got.items():
        gh = gh.float().cpu()
        if name == "model.embed_tokens.weight":
            rows = [spec.audio_start_token, spec.audio_end_token]
            mask = torch.ones(gh.shape[0], dtype=torch.bool)
            mask[rows] = False
            assert gh[mask].abs().max().item() == 0.0
            worst[name] = close(name, gh[rows], z[name + "|rows"])
        elif gh.dim() == 1:
            worst[name] = close(name, gh, z[name + "|full"])
        else:
            r, c = G.sample_index(gh.shape)
            rels = [close(name + " sample", gh[r][:, c], z[name + "|sample"]),
                    close(name + " row sums", gh.double().sum(1), z[name + "|rowsum"]),
                    close(name + " column sums", gh.double().sum(0), z[name + "|colsum"])]
            nrm = float(gh.double().norm())
            assert abs(nrm - float(z[name + "|norm"])) <= 3e-2 * float(z[name + "|norm"]), f"{name}: norm {nrm} vs {float(z[name + '|norm'])}"
            worst[name] = max(rels)
        checked += 1
    assert checked == 9 * spec.num_hidden_layers + 4 and "lm_head.weight" not in got
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"\n[train7b] loss {loss:.5f} (oracle {ref_loss:.5f}); worst gradient errors at 7B width, S = {ids.shape[1]}: "
          + ", ".join(f"{k.replace('model.', '')} {v:.2e}" for k, v in top))


@pytest.mark.parametrize("rows,width,accumulate", [(1000, 4096, False), (1000, 4096, True), (37, 256, False), (5000, 1024, True), (9, 8192, False), (1001, 4096, True), (3, 3000, False)])
def test_rmsnorm_bwd_vs_autograd(rows, width, accumulate):
    """llark_rmsnorm_bwd (rows walked by a fixed grid, dw accumulated in registers) vs torch autograd of LlamaRMSNorm in fp32
    (transformers==4.29.2 modeling_llama.py:LlamaRMSNorm).  fp32 on both sides: 1e-5 relative to the largest entry; dw, a sum
    over `rows` terms accumulated in a different order (atomics), 1e-4."""
    from llark_amd import ops
    g = torch.Generator().manual_seed(rows + width)
    x = torch.randn(rows, width, generator=g)
    w = 1.0 + 0.1 * torch.randn(width, generator=g)
    dy = torch.randn(rows, width, generator=g)
    dx0 = torch.randn(rows, width, generator=g)
    dw0 = torch.randn(width, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5))
    y.backward(dy)
    dx = dx0.clone().cuda()
    dw = dw0.clone().cuda()
    ops.rmsnorm_bwd(x.cuda(), w.cuda(), dy.cuda(), 1e-5, dx, accumulate, dw)
    ref_dx = xr.grad + (dx0 if accumulate else 0.0)
    ref_dw = wr.grad + dw0
    assert (dx.cpu() - ref_dx).abs().max().item() <= 1e-5 * ref_dx.abs().max().item()
    assert (dw.cpu() - ref_dw).abs().max().item() <= 1e-4 * ref_dw.abs().max().item()


def test_grad_norm_clipping_matches_torch():
    """HF Trainer's step clips with torch.nn.utils.clip_grad_norm_(parameters, max_grad_norm) before optimizer.step():
    llark_sumsq_f32 over the flat gradient + the clip coefficient folded into AdamW's gradient scale must give the same norm and
    the same update as torch (fp32 reference on the trainer's own gradients; parameters are bf16: one ulp of slack)."""
    from llark_amd import ops
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup(B=2)
    tr = HipLlamaTrainer(eng, lr=1e-2, weight_decay=0.0, embed_grad_tokens=(spec.audio_start_token, spec.audio_end_token))
    tr.forward_backward(ids.cuda(), segs, labels.cuda())
    tr._finalize_grads()
    ref_norm = tr.flat_grad.double().norm().item()
    assert abs(tr.grad_norm() - ref_norm) <= 1e-6 * ref_norm
    x = torch.randn(1000003, device="cuda")                                  # ragged length: the scalar tail of the kernel
    assert abs(ops.sumsq_f32(x).item() - x.double().pow(2).sum().item()) <= 1e-9 * x.numel()
    # Adam's first update is lr * g / (|g| + eps): blind to a scale of g.  So: one unclipped step, then a clipped one -- the second
    # update mixes the two gradient scales in m and v and moves by a different amount if the coefficient is wrong or missing.
    name = "layers.1.wdown"
    ref_p = torch.nn.Parameter(dict(tr.params)[name].float().clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    ref_p.grad = tr.grads[name].clone()
    opt.step()
    tr.step()
    with torch.no_grad():
        ref_p.copy_(dict(tr.params)[name].float())                           # follow the bf16 rounding of the stored weights
    tr.forward_backward(ids.cuda(), segs, labels.cuda())
    tr._finalize_grads()
    norm2 = tr.flat_grad.double().norm().item()
    max_norm = 0.05 * norm2                                                  # clip coefficient ~0.05
    p_before = dict(tr.params)[name].float().clone()
    ref_p.grad = tr.grads[name].clone() * (max_norm / (norm2 + 1e-6))
    unclipped = torch.nn.Parameter(p_before.clone())                         # what a missing clip would produce
    opt_u = torch.optim.AdamW([unclipped], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    opt_u.load_state_dict(opt.state_dict())
    unclipped.grad = tr.grads[name].clone()
    tr.step(max_grad_norm=max_norm)
    opt.step()
    opt_u.step()
    assert abs(tr.last_grad_norm - norm2) <= 1e-6 * norm2
    p_after = dict(tr.params)[name].float()
    err = (p_after - ref_p.detach().bfloat16().float()).abs().max().item()
    err_if_unclipped = (p_after - unclipped.detach().bfloat16().float()).abs().max().item()
    assert err <= 2 ** -7 * p_before.abs().max().item() + 1e-3, err
    assert err_if_unclipped > 4 * err + 1e-3, (err, err_if_unclipped)         # the test can tell the two apart
    # a norm below the threshold leaves the gradients alone; with last_micro_batch the sums of squares the dW epilogues collected
    # during the backward plus the remaining slices must give the same norm as one pass over the whole buffer (this tiny model's
    # token count, 34, is not a multiple of 64: its dW products take the transposing path and collect nothing -- the 7B-width test
    # below covers the collecting path)
    tr.forward_backward(ids.cuda(), segs, labels.cuda(), last_micro_batch=True)
    tr._finalize_grads()
    full = tr.flat_grad.double().norm().item()
    tr.step(max_grad_norm=1e9)
    assert abs(tr.last_grad_norm - full) <= 1e-9 * full and tr._norm_spans == []
    tr.forward_backward(ids.cuda(), segs, labels.cuda(), last_micro_batch=True)      # partial sums nobody asks for are dropped
    tr.step()
    assert tr._norm_spans == [] and tr.last_grad_norm is None


def test_grad_norm_bookkeeping_invalidated_by_accumulation_and_exchange(monkeypatch):
    """the sums of squares a collecting forward_backward(last_micro_batch=True) leaves behind are only valid
    for the gradients as they stood at its end.  (a) a later accumulating call, (b) a gradient exchange without overlap
    (allreduce_grads(world > 1)) both change the gradients: step(max_grad_norm) must then take the norm of what is in the buffer,
    not reuse the stale partial sums.  64 tokens per micro-batch so that the dW products take the collecting path."""
    from llark_amd import dist as D
    from llark_amd.m2t.engine import HipLlamaEngine, LlamaDims
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    from oracle import llama_ref as LR
    V, F, S, B = 128, 5, 32, 2
    spec = LR.LlamaSpec(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, vocab_size=V,
                        mm_hidden_size=96, audio_start_token=V - 2, audio_end_token=V - 1, audio_patch_token=V - 3)
    w = {k: _bf(v) for k, v in LR.make_weights(spec, seed=0, std=0.08).items()}
    g = torch.Generator().manual_seed(9)
    ids = torch.stack([torch.tensor([1] + torch.randint(3, V - 3, (3,), generator=g).tolist() + [V - 2] + [V - 3] * F + [V - 1]
                                    + torch.randint(3, V - 3, (S - 6 - F,), generator=g).tolist()) for _ in range(B)])
    assert ids.shape == (B, S)
    aud = torch.randn(B, F, 96, generator=g)
    labels = ids.clone()
    labels[:, :12] = -100
    dims = LlamaDims(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, vocab_size=V, mm_hidden_size=96)
    eng = HipLlamaEngine(dims, "cuda", B, 64, precision="bf16")
    eng.load_state_dict(w)
    segs = [(b, 4, aud[b].cuda()) for b in range(B)]
    tr = HipLlamaTrainer(eng, lr=1e-3, weight_decay=0.0, embed_grad_tokens=(V - 2, V - 1))
    # (a) collect, then accumulate once more
    tr.forward_backward(ids.cuda(), segs, labels.cuda(), 0.5, last_micro_batch=True)
    assert tr._norm_spans, "the collecting path did not run: the test does not test anything"
    tr.forward_backward(ids.flip(0).cuda(), [(b, 4, aud[1 - b].cuda()) for b in range(B)], labels.flip(0).cuda(), 0.5)
    assert tr._norm_spans == []
    tr._finalize_grads()
    full = tr.flat_grad.double().norm().item()
    tr.step(max_grad_norm=1e9)
    assert abs(tr.last_grad_norm - full) <= 1e-6 * full, (tr.last_grad_norm, full)
    # (b) collect without overlap, then an exchange over "2 ranks" (stub: the other rank holds 3x this rank's gradient)
    class _Done:
        def wait(self):
            pass

    def fake_all_reduce(t, comm, stage):
        t.mul_(4.0)
        return _Done()

    monkeypatch.setattr(D, "all_reduce_sum_async", fake_all_reduce)
    tr.forward_backward(ids.cuda(), segs, labels.cuda(), 1.0, last_micro_batch=True)
    assert tr._norm_spans
    tr.allreduce_grads(2)
    assert tr._norm_spans == []
    tr._finalize_grads()
    full = tr.flat_grad.double().norm().item() / 2
    tr.step(world=2, max_grad_norm=1e9)
    assert abs(tr.last_grad_norm - full) <= 1e-6 * full, (tr.last_grad_norm, full)


def test_adamw_twins_equals_adamw_plus_packed_twins():
    """llark_adamw_twins (round 6): parameters and moments bit-equal to llark_adamw_clip; wfrag = llark_pack_weight16_frag of the updated
    weight with the q / k head rows in the fused-RoPE order; wtfrag = llark_pack_weight16_frag of its transpose."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    n, k, rope_rows = 384, 256, 256
    p0 = (torch.randn(n, k, generator=g, device="cuda") * 0.05).bfloat16()
    grad = torch.randn(n * k, generator=g, device="cuda") * 1e-2
    m0 = torch.randn(n * k, generator=g, device="cuda") * 1e-3
    v0 = torch.rand(n * k, generator=g, device="cuda") * 1e-5
    sumsq = (grad.double() ** 2).sum().reshape(1)
    for clip in (False, True):
        pa, ma, va = p0.clone(), m0.clone(), v0.clone()
        pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
        kw = dict(grad_sumsq=sumsq, max_grad_norm=0.5) if clip else {}
        ops.adamw(pa.view(-1), grad, ma, va, 1e-3, 0.9, 0.999, 1e-8, 0.01, 3, 0.25, **kw)
        wfrag = torch.full((n * k,), 7.0, dtype=torch.bfloat16, device="cuda")
        wtfrag = torch.full((n * k,), 7.0, dtype=torch.bfloat16, device="cuda")
        ops.adamw_twins(pb, grad, mb, vb, 1e-3, 0.9, 0.999, 1e-8, 0.01, 3, 0.25, wfrag=wfrag, rope_rows=rope_rows, wtfrag=wtfrag, **kw)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
        assert not torch.equal(pa, p0)
        order = ops.rope_qkv_row_order(1, 128).cuda()                     # 384 rows: q and k "heads" permuted, the last 128 rows natural
        assert torch.equal(wfrag, ops.pack_weight16_frag(pb.index_select(0, order), n))
        assert torch.equal(wtfrag, ops.pack_weight16_frag(ops.transposed16(pb), k))
    # only one twin requested; plain row order
    pc, mc, vc = p0.clone(), m0.clone(), v0.clone()
    wf = torch.zeros(n * k, dtype=torch.bfloat16, device="cuda")
    ops.adamw_twins(pc, grad, mc, vc, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, wfrag=wf)
    assert torch.equal(wf, ops.pack_weight16_frag(pc, n))


@pytest.mark.parametrize("m,H,I", [(300, 256, 384), (4096, 4096, 11008), (4100, 4096, 11008)])
def test_swiglu_train_epilogues_match_the_two_launch_path(m, H, I):
    """llark_gemm16_fragw_swiglu_train: mode 0 = gate|up product + SwiGLU (act from the fp32 accumulators like llark_swiglu_fwd; gate | up
    left as bf16); mode 1 = d(act) product + SwiGLU backward on those bf16 values.  Compared with the separate kernels run on the same
    bf16-rounded gate | up: only the hardware exp2 / rcp of the epilogue (<= 1e-6 relative) and one bf16 rounding differ."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(8)
    ws = 0.08 if H <= 256 else 0.02                                     # (the 7B shapes at a 7B-like weight scale; ragged row tile at m = 4100)
    x = torch.randn(m, H, generator=g, device="cuda").bfloat16()
    wgu = (torch.randn(2 * I, H, generator=g, device="cuda") * ws).bfloat16()
    wdown = (torch.randn(H, I, generator=g, device="cuda") * ws).bfloat16()
    act = torch.empty(m, I, dtype=torch.bfloat16, device="cuda")
    gu16 = torch.empty(m, 2 * I, dtype=torch.bfloat16, device="cuda")
    assert ops.gemm16_fragw_swiglu_train(0, x, ops.pack_weight16_frag(wgu, 2 * I), 2 * I, H, act, gu16)
    gu = torch.empty(m, 2 * I, device="cuda")
    ops.gemm16(x, None, wgu, None, 2 * I, ops.EPI_F32, c=gu)
    act_ref = torch.empty_like(act)
    ops.swiglu_fwd(gu, act_ref)
    assert torch.equal(gu16, gu.bfloat16())
    assert (act.float() - act_ref.float()).abs().max().item() <= 2 ** -7 * act_ref.float().abs().max().item()
    dh = torch.randn(m, H, generator=g, device="cuda").bfloat16()
    dgu = torch.empty(m, 2 * I, dtype=torch.bfloat16, device="cuda")
    wdT = ops.transposed16(wdown)                                       # [I][H]
    assert ops.gemm16_fragw_swiglu_train(1, dh, ops.pack_weight16_frag(wdT, I), I, H, dgu, gu16)
    dact = torch.empty(m, I, device="cuda")
    ops.gemm16_t(dh, wdown, m, I, H, False, True, dact)
    dgu_ref = torch.empty_like(dgu)
    ops.swiglu_bwd(gu16.float(), dact, dgu_ref)
    err = (dgu.float() - dgu_ref.float()).abs().max().item()
    assert err <= 2 ** -6 * dgu_ref.float().abs().max().item(), err
    assert dgu_ref.float().abs().max().item() > 0


def _setup_long(S=72, B=2, layers=2):
    """Like _setup with sequences long enough for the fragment-major paths (>= 129 rows) and the fused-RoPE epilogue (S >= 32)."""
    from llark_amd.m2t.engine import HipLlamaEngine, LlamaDims
    from oracle import llama_ref as LR
    V = 128
    spec = LR.LlamaSpec(hidden_size=256, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=2, vocab_size=V,
                        mm_hidden_size=96, audio_start_token=V - 2, audio_end_token=V - 1, audio_patch_token=V - 3)
    w = {k: _bf(v) for k, v in LR.make_weights(spec, seed=0, std=0.08).items()}
    g = torch.Generator().manual_seed(6)
    F = 5
    ids = torch.stack([torch.tensor([1] + torch.randint(3, V - 3, (2 + b,), generator=g).tolist() + [V - 2] + [V - 3] * F + [V - 1]
                                    + torch.randint(3, V - 3, (S - F - 5 - b,), generator=g).tolist()) for b in range(B)])
    assert ids.shape == (B, S)
    aud = torch.randn(B, F, 96, generator=g)
    labels = ids.clone()
    labels[:, :10] = -100
    dims = LlamaDims(hidden_size=256, intermediate_size=512, num_hidden_layers=layers, num_attention_heads=2, vocab_size=V, mm_hidden_size=96)
    eng = HipLlamaEngine(dims, "cuda", B, 128, precision="bf16")
    eng.load_state_dict(w)
    segs = [(b, int((ids[b] == V - 2).nonzero()[0, 0]), aud[b].cuda()) for b in range(B)]
    return spec, w, ids, aud, labels, eng, segs


def test_twin_paths_gradients_match_autograd_and_survive_an_optimizer_step():
    """Round 6: with >= 129 rows and S >= 32 every product of the step runs on the optimizer-maintained twins -- forward on W fragment-major
    (q|k|v with RoPE in the epilogue, gate|up with SwiGLU in the epilogue), dX on W^T fragment-major (down_proj's with the SwiGLU backward
    in the epilogue).  (1) gradients vs torch autograd of the oracle, same bars as the small-shape test; (2) after one AdamW step the twins
    ARE the packed updated weights, and the next micro-batch's gradients equal those of a fresh trainer built on the updated weights."""
    from llark_amd import ops
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup_long()
    toks = (spec.audio_start_token, spec.audio_end_token)
    tr = HipLlamaTrainer(eng, lr=1e-2, weight_decay=0.0, embed_grad_tokens=toks)
    assert tr.twins and tr.rope_fused and tr.swiglu_fused and len(tr.twins) == 4 * spec.num_hidden_layers
    loss = tr.forward_backward(ids.cuda(), segs, labels.cuda()).item()
    ref_loss, ref = _oracle_grads(spec, w, ids, aud, labels)
    assert abs(loss - ref_loss) <= 5e-3 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    for name, gh in tr.export_grads_hf().items():
        r = ref[name]
        gh = gh.float().cpu()
        if name == "model.embed_tokens.weight":
            rows = [spec.audio_start_token, spec.audio_end_token]
            gh, r = gh[rows], r[rows]
        rel = ((gh - r).norm() / (r.norm() + 1e-30)).item()
        cos = torch.nn.functional.cosine_similarity(gh.flatten(), r.flatten(), dim=0).item()
        assert np.isfinite(rel) and rel <= 3e-2 and cos >= 0.999, f"{name}: rel {rel:.3e} cos {cos:.5f}"
    tr.step(max_grad_norm=1.0)
    H = spec.hidden_size
    order = ops.rope_qkv_row_order(spec.num_attention_heads, 128).cuda()
    for name, (wfrag, wtfrag, rope_rows) in tr.twins.items():
        p = dict(tr.params)[name]
        n, k = p.shape
        src = p.index_select(0, order) if rope_rows else p
        assert rope_rows == (2 * H if name.endswith("wqkv") else 0)
        assert torch.equal(wfrag, ops.pack_weight16_frag(src, n)), name
        assert torch.equal(wtfrag, ops.pack_weight16_frag(ops.transposed16(p), k)), name
    tr.forward_backward(ids.cuda(), segs, labels.cuda())
    g_next = tr.flat_grad.clone()
    tr2 = HipLlamaTrainer(eng, lr=1e-2, weight_decay=0.0, embed_grad_tokens=toks)       # twins packed from the updated weights
    tr2.forward_backward(ids.cuda(), segs, labels.cuda())
    for name, prm in tr.params:
        if prm.dim() == 2:
            assert torch.equal(tr.grads[name], tr2.grads[name]), name


def test_twin_paths_agree_with_the_untwinned_trainer(monkeypatch):
    """The same micro-batch through the round-5 paths (LLARK_TRAIN_TWINS=0: generic forward kernels, llark_gemm16_t for dX, separate RoPE /
    SwiGLU launches with fp32 gate | up) and through the twins: gradients agree to bf16-flow noise."""
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup_long()
    toks = (spec.audio_start_token, spec.audio_end_token)
    tr = HipLlamaTrainer(eng, embed_grad_tokens=toks)
    tr.forward_backward(ids.cuda(), segs, labels.cuda())
    monkeypatch.setenv("LLARK_TRAIN_TWINS", "0")
    tr0 = HipLlamaTrainer(eng, embed_grad_tokens=toks)
    assert not tr0.twins
    tr0.forward_backward(ids.cuda(), segs, labels.cuda())
    for name, prm in tr.params:
        a, b = tr.grads[name].float(), tr0.grads[name].float()
        if b.norm().item() == 0.0:
            assert a.norm().item() == 0.0, name
            continue
        rel = ((a - b).norm() / b.norm()).item()
        assert rel <= 2e-2, f"{name}: {rel:.3e}"


def test_rope_qkv_train_epilogue_writes_v_row_major():
    """llark_gemm16_fragw_rope_qkv_train: q, the K cache and the V^T cache exactly as llark_gemm16_fragw_rope_qkv, plus V row-major
    [b][nh][s][128] == the transpose of what went into the V^T cache."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(21)
    B, S, nh, hd, smax, H = 2, 72, 2, 128, 128, 256
    bf = torch.bfloat16
    x = torch.randn(B * S, H, generator=g, device="cuda").to(bf)
    w = (torch.randn(3 * H, H, generator=g, device="cuda") * 0.06).to(bf)
    wfrag = ops.pack_weight16_frag(w.index_select(0, ops.rope_qkv_row_order(nh, hd).cuda()), 3 * H)
    half = hd // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, half, device="cuda", dtype=torch.float32) / half))
    ang = torch.arange(0, 256, device="cuda", dtype=torch.float32)[:, None] * inv[None, :]
    cos_t, sin_t = ang.cos().contiguous(), ang.sin().contiguous()
    outs = []
    for train in (False, True):
        q = torch.zeros(B, nh, S, hd, dtype=bf, device="cuda")
        kc = torch.zeros(B, nh, smax, hd, dtype=bf, device="cuda")
        vt = torch.zeros(B, nh, hd, smax, dtype=bf, device="cuda")
        v_rm = torch.full((B * nh, S, hd), 9.0, dtype=bf, device="cuda")
        if train:
            ops.gemm16_fragw_rope_qkv_train(x, wfrag, H, B, S, nh, 0, cos_t, sin_t, q, kc, vt, v_rm)
        else:
            ops.gemm16_fragw_rope_qkv(x, None, wfrag, H, B, S, nh, 0, cos_t, sin_t, q, kc, vt)
        outs.append((q, kc, vt, v_rm))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    vt, v_rm = outs[1][2], outs[1][3]
    assert torch.equal(v_rm.view(B, nh, S, hd), vt[:, :, :, :S].transpose(2, 3))
    assert v_rm.float().abs().max().item() > 0


@pytest.mark.parametrize("rows,width,accumulate", [(300, 4096, True), (64, 256, False), (130, 1024, True)])
def test_rmsnorm_bwd_bf16_copy_equals_split16_of_its_dx(rows, width, accumulate):
    """llark_rmsnorm_bwd_out16 (round 6): dx and the gain gradient exactly as llark_rmsnorm_bwd, plus bf16(dx) == llark_split16's hi plane."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(rows + width)
    x = torch.randn(rows, width, generator=g, device="cuda")
    w = 1 + 0.1 * torch.randn(width, generator=g, device="cuda")
    dy = torch.randn(rows, width, generator=g, device="cuda") * 0.3
    dx0 = torch.randn(rows, width, generator=g, device="cuda")
    outs = []
    for fused in (False, True):
        dx, dw = dx0.clone(), torch.zeros(width, device="cuda")
        d16 = torch.full((rows, width + 64), 5.0, dtype=torch.bfloat16, device="cuda")[:, :width] if fused else None
        ops.rmsnorm_bwd(x, w, dy, 1e-5, dx, accumulate, dw, dx16=d16)
        outs.append((dx, dw, d16))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-4)          # per-workgroup partials meet in fp32 atomics: order varies run to run
    hi, _ = ops.split16(outs[1][0], torch.bfloat16, want_lo=False, kmult=64)
    assert torch.equal(outs[1][2], hi[:, :width])


def test_fused_accumulation_equals_separate_micro_batches():
    """``forward_backward(..., loss_groups=G)``: G of the recipe's micro-batches (train_llark.sh:26-27: per_device_train_batch_size 2 x
    gradient_accumulation_steps 4) in ONE pass with the loss normalised per group = the gradients G separate calls accumulate (same
    products, fp32 sums in a different order: 2e-3 relative per tensor against ~3e-2 for either against autograd), the same mean loss;
    label counts differ between the groups, so a global token mean would NOT pass."""
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    spec, w, ids, aud, labels, eng, segs = _setup_long(S=72, B=4)
    labels[2:, :40] = -100                                   # the second group has fewer label tokens than the first
    toks = (spec.audio_start_token, spec.audio_end_token)
    tr = HipLlamaTrainer(eng, lr=1e-2, weight_decay=0.0, embed_grad_tokens=toks)
    losses = []
    for gi in range(2):
        sl = slice(2 * gi, 2 * gi + 2)
        sg = [(b - 2 * gi, st, a) for (b, st, a) in segs[2 * gi: 2 * gi + 2]]
        losses.append(tr.forward_backward(ids[sl].cuda(), sg, labels[sl].cuda(), 0.5).item())
    sep = {k: v.float().clone() for k, v in tr.export_grads_hf().items()}
    tr.zero_grad()
    loss = tr.forward_backward(ids.cuda(), segs, labels.cuda(), 0.5, loss_groups=2).item()
    assert abs(loss - 0.5 * (losses[0] + losses[1])) <= 1e-5 * abs(loss)
    for name, gf in tr.export_grads_hf().items():
        r = sep[name]
        rel = ((gf.float() - r).norm() / (r.norm() + 1e-30)).item()
        assert np.isfinite(rel) and rel <= 2e-3, f"{name}: rel {rel:.3e}"
    tr.zero_grad()
    glob = tr.forward_backward(ids.cuda(), segs, labels.cuda(), 1.0)          # one global token mean: a different gradient
    rel = max(((gf.float() - sep[name]).norm() / (sep[name].norm() + 1e-30)).item() for name, gf in tr.export_grads_hf().items())
    assert rel > 1e-2


@pytest.mark.parametrize("n,k,rope_heads", [(12288, 4096, 32), (4096, 4096, 0), (22016, 4096, 0), (4096, 11008, 0)])
def test_adamw_twins_at_the_7b_weight_shapes(n, k, rope_heads):
    """The same equalities as the small-shape test on the four weight shapes of a Llama-2-7B layer (q|k|v with its RoPE row order over 2 x 32
    heads, o_proj, gate|up, down_proj): 50-90 M elements each -- index arithmetic and tile edges at the sizes the step runs."""
    from llark_amd import ops
    g = torch.Generator(device="cuda").manual_seed(n + k)
    p0 = (torch.randn(n, k, generator=g, device="cuda") * 0.02).bfloat16()
    grad = torch.randn(n * k, generator=g, device="cuda") * 1e-3
    m0 = torch.randn(n * k, generator=g, device="cuda") * 1e-4
    v0 = torch.rand(n * k, generator=g, device="cuda") * 1e-7
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    ops.adamw(pa.view(-1), grad, ma, va, 5e-5, 0.9, 0.999, 1e-8, 0.0, 2, 0.25)
    wfrag = torch.empty((n * k,), dtype=torch.bfloat16, device="cuda")
    wtfrag = torch.empty((n * k,), dtype=torch.bfloat16, device="cuda")
    rope_rows = 2 * rope_heads * 128
    ops.adamw_twins(pb, grad, mb, vb, 5e-5, 0.9, 0.999, 1e-8, 0.0, 2, 0.25, wfrag=wfrag, rope_rows=rope_rows, wtfrag=wtfrag)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and not torch.equal(pa, p0)
    src = pb.index_select(0, ops.rope_qkv_row_order(rope_heads, 128).cuda()) if rope_heads else pb
    assert torch.equal(wfrag, ops.pack_weight16_frag(src, n))
    assert torch.equal(wtfrag, ops.pack_weight16_frag(ops.transposed16(pb), k))


def test_round6_paths_agree_with_round5_paths_at_7b_width(monkeypatch):
    """The whole micro-batch at the width and grid sizes the step runs -- hidden 4096, 32 heads, intermediate 11008, one decoder layer, 2 x 1024
    tokens (the attention kernels on their paired-block grids: 64 heads x 16 blocks) -- through round 6's paths (twins, fused RoPE / SwiGLU /
    attention glue, dW on the DMA loop) and through round 5's (every LLARK_TRAIN_* switch off): the gradients agree to the bf16-flow noise of
    the small-width test.  Complements the autograd fixture above (one sequence of 1024: unpaired attention grids)."""
    from llark_amd.m2t.engine import HipLlamaEngine, LlamaDims
    from llark_amd.m2t.train_engine import HipLlamaTrainer
    V, B, S, F = 32004, 2, 1024, 25
    dims = LlamaDims(num_hidden_layers=1, vocab_size=V)
    assert (dims.hidden_size, dims.intermediate_size, dims.num_attention_heads) == (4096, 11008, 32)
    eng = HipLlamaEngine(dims, "cuda", B, S, precision="bf16")
    g = torch.Generator(device="cuda").manual_seed(5)
    H, I = dims.hidden_size, dims.intermediate_size

    def n(*shape, std=0.02):
        return (torch.randn(*shape, generator=g, device="cuda") * std).bfloat16()
    ones = torch.ones(H, device="cuda")
    eng.set_layer(0, n(H, H), n(H, H), n(H, H), n(H, H), n(I, H), n(I, H), n(H, I), ones, ones)
    eng.set_globals(n(V, H), ones, n(V, H), n(H, dims.mm_hidden_size), torch.zeros(H, device="cuda"))
    cg = torch.Generator().manual_seed(6)
    ids = torch.randint(3, 32000, (B, S), generator=cg)
    labels = ids.clone()
    labels[:, :40] = -100
    emb = torch.randn(B, F, dims.mm_hidden_size, generator=cg).cuda()
    segs = [(b, 1, emb[b]) for b in range(B)]
    tr = HipLlamaTrainer(eng, embed_grad_tokens=[32001, 32002])
    assert tr.twins and tr.rope_fused and tr.swiglu_fused and tr.dw_fragw and tr.attn_glue_fused
    la = tr.forward_backward(ids.cuda(), segs, labels.cuda()).item()
    for k in ("LLARK_TRAIN_TWINS", "LLARK_TRAIN_DW_FRAGW", "LLARK_TRAIN_ATTN_GLUE_FUSED", "LLARK_TRAIN_NORM_BWD_OUT16"):
        monkeypatch.setenv(k, "0")
    tr0 = HipLlamaTrainer(eng, embed_grad_tokens=[32001, 32002])
    assert not tr0.twins and not tr0.dw_fragw and not tr0.attn_glue_fused
    lb = tr0.forward_backward(ids.cuda(), segs, labels.cuda()).item()
    assert abs(la - lb) <= 2e-3 * abs(lb), (la, lb)
    for name, prm in tr.params:
        a, b = tr.grads[name].float(), tr0.grads[name].float()
        if b.norm().item() == 0.0:
            assert a.norm().item() == 0.0, name
            continue
        rel = ((a - b).norm() / b.norm()).item()
        assert rel <= 2e-2, f"{name}: {rel:.3e}"
