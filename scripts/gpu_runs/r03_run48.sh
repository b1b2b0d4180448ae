#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_attn_bwd_gpu.py tests/test_train_gpu.py tests/test_mpt_gpu.py -q 2>&1 | tail -6 ) > gpurun_out/r03_run48_tests.txt; cat gpurun_out/r03_run48_tests.txt
timeout 900 python bench.py --stages train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_train_v7.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_v7.json'));print('4x512:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'],d['kernel_ms'])"
timeout 900 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_v7.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_v7.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'],d['kernel_ms'])"
timeout 900 python bench.py --stages mpt-train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_mpt_train_v7.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_mpt_train_v7.json'));print('mpt-train:',d['ms_per_step'],d['value'])"
