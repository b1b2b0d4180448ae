#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gemm_tn_gpu.py tests/test_train_gpu.py -q -x 2>&1 | tail -6 ) > gpurun_out/r03_run52_tests.txt; cat gpurun_out/r03_run52_tests.txt
timeout 600 python scripts/bench_gemm_train.py 11,12,200 4096 2>&1 | grep -v amdgpu.ids | grep "dX\|dW" | grep "variant 12\|variant 200" | tee gpurun_out/r03_gemm_tn_wide_m4096.txt
timeout 900 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_v9.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_v9.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'],d['kernel_ms'])"
