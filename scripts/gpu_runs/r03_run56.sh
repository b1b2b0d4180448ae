#!/bin/bash
# dX products: W^T + fragment twins from the second use (default) vs llark_gemm16_t on W for every micro-batch
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
for u in 1 1000; do
LLARK_TRAIN_DX_DIRECT_USES=$u timeout 400 python bench.py --stages train --no-cpu-baseline --batch 8 --micro-batch 2 --train-seq 2048 2>&1 | tail -1 > gpurun_out/r03_bench_train_2x2048_dx$u.json; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_train_2x2048_dx$u.json'));print('direct uses $u: 2x2048x4:',d['ms_per_step'],d['value'],d.get('mfu'),d['peak_hbm_gb'])"
done
