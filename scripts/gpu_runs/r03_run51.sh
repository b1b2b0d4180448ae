#!/bin/bash
# end-of-round validation: full GPU suite, smoke, default bench (+ kernel trace of the same command), train / generate / llama lines
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r03_gpu_suite_final.txt; cat gpurun_out/r03_gpu_suite_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r03_bench_e2e_final.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_e2e_final.json'));print('e2e:',d['ms_per_step'],d['value'],d['roofline']['frac'],d.get('alt_prior_precision',{}).get('value'),d['kernel_ms'])"
timeout 900 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_final.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_final.json'));print('2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'])"
timeout 900 python bench.py --stages train --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_train_final.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_final.json'));print('4x512:',d['ms_per_step'],d['value'],d.get('mfu'))"
timeout 900 python bench.py --stages generate --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_generate_final.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_generate_final.json'));print('generate:',d['ms_per_step'],d['value'])"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_trace_e2e -o e2e -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt-precision --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocprof_summary.py $(find gpurun_out/r03_trace_e2e -name '*.db' | head -1) gpurun_out/r03_e2e_final_kernel_stats.txt; head -12 gpurun_out/r03_e2e_final_kernel_stats.txt | cut -c1-140; rm -rf gpurun_out/r03_trace_e2e
