#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_llama_gpu.py tests/test_fulldepth_gpu.py tests/test_infer_driver.py -q 2>&1 | tail -8 ) > gpurun_out/r03_run45_tests.txt; cat gpurun_out/r03_run45_tests.txt
echo "== decode (split-KV attention)"; timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
echo "== LLARK_DECODE_SPLIT_KV=0"; LLARK_DECODE_SPLIT_KV=0 timeout 300 python scripts/bench_decode.py split 2>&1 | grep "decode" | grep "B=1"
