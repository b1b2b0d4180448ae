#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_train_gpu.py tests/test_mpt_gpu.py tests/test_gemm_tn_gpu.py -q 2>&1 | tail -8 ) > gpurun_out/r03_run28_tests.txt; cat gpurun_out/r03_run28_tests.txt
timeout 900 python bench.py --stages train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_train_v5.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_v5.json'));print('4x512:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'],d['kernel_ms'])"
timeout 900 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_v5.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_v5.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d.get('allreduce_exposed_ms'),d['peak_hbm_gb'],d['kernel_ms'])"
timeout 900 python bench.py --stages mpt-train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_mpt_train_v5.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_mpt_train_v5.json'));print('mpt-train:',d['ms_per_step'],d['value'])"
