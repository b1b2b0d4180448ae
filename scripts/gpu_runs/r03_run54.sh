#!/bin/bash
# validation of the shipped library after the wide gemm_tn tile, DQ_NQ=1 and the two-wave rmsnorm backward: changed-kernel tests first,
# smoke, the train / default bench lines, then as much of the full GPU suite as the budget allows
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_train_gpu.py tests/test_attn_bwd_gpu.py tests/test_gemm_tn_gpu.py tests/test_mpt_gpu.py -q -m gpu -x 2>&1 | tail -4 ) > gpurun_out/r03_changed_tests.txt; cat gpurun_out/r03_changed_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python scripts/bench_attn.py 2>&1 | tail -8 > gpurun_out/r03_bench_attn_v2.txt; cat gpurun_out/r03_bench_attn_v2.txt
timeout 400 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_v7.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_v7.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'])"
timeout 400 python bench.py --stages train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_train_v7.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_v7.json'));print('4x512:',d['ms_per_step'],d['value'],d.get('mfu'))"
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/r03_bench_e2e_v7.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_e2e_v7.json'));print('e2e:',d['ms_per_step'],d['value'],d['roofline']['frac'],d.get('alt_prior_precision',{}).get('value'),d['kernel_ms'])"
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_train_gpu.py --deselect tests/test_attn_bwd_gpu.py --deselect tests/test_gemm_tn_gpu.py --deselect tests/test_mpt_gpu.py 2>&1 | tee gpurun_out/r03_rest_tests_full.txt | tail -4 > gpurun_out/r03_rest_tests.txt; cat gpurun_out/r03_rest_tests.txt
