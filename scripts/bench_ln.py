#!/usr/bin/env python3
"""LayerNorm -> (fp16 hi, E4M3 low) planes at the prior's shape (8 clips x 8192 tokens x 4800), run on the GPU box."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from llark_amd import ops  # noqa: E402

rows, W = 8 * 8192, 4800
g = torch.Generator(device="cuda").manual_seed(0)
xs = [torch.randn(rows, W, generator=g, device="cuda") for _ in range(3)]      # rotate: 1.26 GB each, nothing stays cached
gam, bet = torch.ones(W, device="cuda"), torch.zeros(W, device="cuda")
hi = torch.zeros(rows, 4800, dtype=torch.float16, device="cuda")
lo = torch.zeros(rows, 4800, dtype=torch.uint8, device="cuda")
for i in range(3):
    ops.layernorm_split_lo8(xs[i], gam, bet, 1e-5, hi, lo)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
n = 30
for i in range(n):
    ops.layernorm_split_lo8(xs[i % 3], gam, bet, 1e-5, hi, lo)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f"layernorm_split_lo8: {ms * 1e3:.1f} us  ({rows * W * 7 / ms / 1e9:.2f} TB/s of 4 B read + 3 B written per element)")
