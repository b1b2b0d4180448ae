#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_attn_bwd_gpu.py -x -q 2>&1 | tail -15 ) > gpurun_out/r03_run14_attn.txt; cat gpurun_out/r03_run14_attn.txt
( timeout 900 python -m pytest tests/test_train_gpu.py tests/test_llama_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/r03_run14_train.txt; cat gpurun_out/r03_run14_train.txt
timeout 600 python bench.py --stages train --steps 3 --warmup 1 2>&1 | grep "^{" > gpurun_out/r03_run14_bench_train.json; cut -c1-1500 gpurun_out/r03_run14_bench_train.json
