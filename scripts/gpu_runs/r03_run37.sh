#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
echo "== tests, fused norm A"; ( LLARK_DECODE_FUSE_NORM_A=1 timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_fulldepth_gpu.py -q -k "llama or generate or decode or logits or greedy" 2>&1 | tail -5 )
echo "== default"; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
echo "== LLARK_DECODE_FUSE_NORM_A=1"; LLARK_DECODE_FUSE_NORM_A=1 timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
echo "== bf16 default"; timeout 300 python scripts/bench_decode.py bf16 2>&1 | grep "decode" | grep "B=1" | head -2
echo "== bf16 LLARK_DECODE_FUSE_NORM_A=1"; LLARK_DECODE_FUSE_NORM_A=1 timeout 300 python scripts/bench_decode.py bf16 2>&1 | grep "decode" | grep "B=1" | head -2
