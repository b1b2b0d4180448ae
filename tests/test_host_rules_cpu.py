"""CPU: host-side dispatch rules added in round 3 -- which decode shapes go to the LDS-DMA streaming Linear (must mirror
gemv_shape_ok in csrc/gemv_dma.hip: the Python side decides BEFORE the call so that a recorded launch list never holds a call the
library would refuse), and the training defaults that mirror HF Trainer (gradient clipping)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _c_shape_ok(split, m, kp):
    """gemv_shape_ok of csrc/gemv_dma.hip restated from its source constants (parsed, so a change there fails this test)."""
    src = (ROOT / "llark_amd" / "csrc" / "gemv_dma.hip").read_text()
    chunk = int(re.search(r"constexpr int GV_CHUNK = (\d+);", src).group(1))
    maxch = int(re.search(r"constexpr int GV_MAXCH = (\d+);", src).group(1))
    xlds = eval(re.search(r"constexpr int GV_XLDS_MAX = ([^;]+);", src).group(1))
    if m < 1 or m > 4 or kp % 8 or kp > maxch * chunk:
        return False
    if kp <= chunk and m <= 2:
        return True
    mm = m if m <= 2 else 4
    return mm * (2 if split else 1) * ((kp + chunk - 1) // chunk) * chunk * 2 <= xlds


def test_python_rule_matches_the_kernel_source():
    from llark_amd import ops
    for split in (False, True):
        for m in range(0, 7):
            for kp in (8, 256, 4096, 4104, 8192, 11008, 12288, 12296, 16384, 4100):
                assert ops._gemv_dma_takes(split, m, kp) == _c_shape_ok(split, m, kp), (split, m, kp)


def test_streaming_rule_thresholds():
    from llark_amd import ops
    big = ops.GEMV_DMA_MIN_BYTES
    assert big == 64 * 1000 * 1000
    # Llama-2-7B decode shapes at B = 1: q/k/v, gate/up, down, lm_head stream; o_proj (33.5 MB) stays on the MFMA kernel
    for n, k, want in ((12288, 4096, True), (22016, 4096, True), (4096, 11008, True), (32004, 4096, True), (4096, 4096, False)):
        assert (n * k * 2 >= big) == want
    assert ops.gemv_dma_rmsnorm_takes(1, 12288, 4096) and ops.gemv_dma_rmsnorm_takes(1, 32004, 4096)
    assert not ops.gemv_dma_rmsnorm_takes(2, 12288, 4096)          # the fused-norm form holds ONE row in registers
    assert not ops.gemv_dma_rmsnorm_takes(1, 4096, 11008)          # ... of at most 4096
    assert not ops.gemv_dma_rmsnorm_takes(1, 6144, 2048)           # MPT-1B widths: below the streaming threshold


def test_training_defaults_mirror_hf_trainer():
    from llark_amd.m2t.train import TrainConfig, lr_at
    cfg = TrainConfig()
    assert cfg.max_grad_norm == 1.0                                 # transformers TrainingArguments default; train_llark.sh does not set it
    assert cfg.gradient_accumulation_steps == 4 and cfg.learning_rate == 5e-5 and cfg.lr_scheduler_type == "cosine"
    assert lr_at(0, cfg) == 0.0 and abs(lr_at(3000, cfg) - 5e-5) < 1e-12
