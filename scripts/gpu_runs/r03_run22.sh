#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 900 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_flash3.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_flash3.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d.get('allreduce_exposed_ms'),d['peak_hbm_gb'],d['kernel_ms'])"
timeout 900 python bench.py --stages train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_train_flash3.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_flash3.json'));print('4x512:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'],d['kernel_ms'])"
timeout 900 python bench.py --stages mpt-train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_mpt_train.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_mpt_train.json'));print('mpt-train:',d['ms_per_step'],d['value'],d.get('mfu'),d['config']['workload'][:120])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_flash -o tr -- python $GRAFT_REPO_ROOT/bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 --steps 1 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_flash -name '*.db' | head -1) gpurun_out/r03_train_2x2048_flash_kernel_stats.txt; head -24 gpurun_out/r03_train_2x2048_flash_kernel_stats.txt | cut -c1-175; rm -rf gpurun_out/r03_trace_flash
