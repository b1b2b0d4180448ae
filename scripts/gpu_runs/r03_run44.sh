#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
( timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r03_gpu_suite_run4.txt; cat gpurun_out/r03_gpu_suite_run4.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r03_bench_e2e_v6.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_e2e_v6.json'));print('e2e:',d['ms_per_step'],d['value'],d['roofline']['frac'],d.get('alt_prior_precision',{}).get('value'),d['kernel_ms'])"
