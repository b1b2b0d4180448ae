#!/bin/bash
# default bench line (without the CPU baseline leg) on the round's last library
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout 75 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03_bench_e2e_v9.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_e2e_v9.json'));print('e2e:',d['ms_per_step'],d['value'],d['roofline']['frac'],d.get('alt_prior_precision',{}).get('value'),d['kernel_ms'])"
