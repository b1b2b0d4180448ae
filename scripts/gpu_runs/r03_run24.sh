#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r03_gpu_suite_run2.txt; cat gpurun_out/r03_gpu_suite_run2.txt
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r03_bench_e2e_v4.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_e2e_v4.json'));print('e2e:',d['ms_per_step'],d['value'],d['roofline'],d.get('alt_prior_precision'),d['kernel_ms'])"
