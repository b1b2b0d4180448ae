#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_attn_bwd_gpu.py -q 2>&1 | tail -15 ) > gpurun_out/r03_run15_attn.txt; cat gpurun_out/r03_run15_attn.txt
timeout 900 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_flash.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_flash.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d.get('allreduce_exposed_ms'),d['peak_hbm_gb'],d['kernel_ms'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_flash -o tr -- python $GRAFT_REPO_ROOT/bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 --steps 1 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_flash -name '*.db' | head -1) gpurun_out/r03_train_2x2048_flash_kernel_stats.txt; head -28 gpurun_out/r03_train_2x2048_flash_kernel_stats.txt | cut -c1-175; rm -rf gpurun_out/r03_trace_flash
