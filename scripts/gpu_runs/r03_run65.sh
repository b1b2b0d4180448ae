#!/bin/bash
# last validation of the round on the shipped library (attention LDS layout + pipeline, RMSNorm backward 4 waves per row): tests of every
# changed kernel, smoke, train lines, then as much of the remaining suite as the GPU budget allows
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 150 python -m pytest tests/test_train_gpu.py tests/test_attn_bwd_gpu.py tests/test_gemm_tn_gpu.py tests/test_mpt_gpu.py tests/test_llama_gpu.py -q -m gpu -x 2>&1 | tail -3 ) > gpurun_out/r03_changed_tests_v2.txt; cat gpurun_out/r03_changed_tests_v2.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 120 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_v9.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_v9.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'])"
timeout 100 python bench.py --stages train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_train_v9.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_v9.json'));print('4x512:',d['ms_per_step'],d['value'],d.get('mfu'))"
timeout 100 python -m pytest tests -q -m gpu -x --deselect tests/test_train_gpu.py --deselect tests/test_attn_bwd_gpu.py --deselect tests/test_gemm_tn_gpu.py --deselect tests/test_mpt_gpu.py --deselect tests/test_llama_gpu.py 2>&1 | tail -4 > gpurun_out/r03_rest_tests_v2.txt; cat gpurun_out/r03_rest_tests_v2.txt
