#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gemm_tn_gpu.py tests/test_train_gpu.py tests/test_mpt_gpu.py -q -x 2>&1 | tail -6 ) > gpurun_out/r03_run50_tests.txt; cat gpurun_out/r03_run50_tests.txt
timeout 900 python bench.py --stages train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_train_v8.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_v8.json'));print('4x512:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'],d['kernel_ms'])"
