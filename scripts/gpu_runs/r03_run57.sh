#!/bin/bash
# dX on W as stored for every micro-batch (new default): trainer tests, train lines, kernel trace of the recipe-length step
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -3 ) > gpurun_out/r03_train_tests_v8.txt; cat gpurun_out/r03_train_tests_v8.txt
timeout 400 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_v8.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_v8.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'])"
timeout 400 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 --grad-checkpoint 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_ckpt_v8.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_ckpt_v8.json'));print('2x2048x4 ckpt:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'])"
timeout 400 python bench.py --stages train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_train_v8.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_v8.json'));print('4x512:',d['ms_per_step'],d['value'],d.get('mfu'))"
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_train -o tr -- python $GRAFT_REPO_ROOT/bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_train -name '*.db' | head -1) gpurun_out/r03_train_2x2048_v8_kernel_stats.txt; head -24 gpurun_out/r03_train_2x2048_v8_kernel_stats.txt | cut -c1-150; rm -rf gpurun_out/r03_trace_train
