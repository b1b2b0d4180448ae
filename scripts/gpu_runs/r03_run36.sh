#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
echo "== default"; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
echo "== LLARK_DECODE_FUSE_NORM_A=1"; LLARK_DECODE_FUSE_NORM_A=1 timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
echo "== LLARK_DECODE_GRAPH=1"; LLARK_DECODE_GRAPH=1 timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
