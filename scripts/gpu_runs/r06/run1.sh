#!/bin/bash
# Round 6, first GPU call: parity of the new training kernels (two-stage dW variants, AdamW twins, SwiGLU train epilogues, the trainer's
# twin paths), tile-variant microbench at the recipe's micro-batch (M = 4096), train stage A/B at the recipe shape (8 x 2048 = micro 2 x 4).
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
{
echo "== tests"
timeout 900 python -m pytest tests/test_gemm_tn_gpu.py tests/test_train_gpu.py -q -x --tb=short -p no:cacheprovider 2>&1 | tail -15
echo "== gemm variants M=4096"
timeout 600 python scripts/bench_gemm_train.py 210,220,102 4096 2>&1 | grep -v "^$" | grep -E "dX|dW|sum over"
echo "== train stage, recipe shape"
for tw in 0 1; do
  echo "-- LLARK_TRAIN_TWINS=$tw"
  LLARK_TRAIN_TWINS=$tw timeout 900 python bench.py --stages train --batch 8 --micro-batch 2 --train-seq 2048 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'mfu', d.get('mfu') or d.get('roofline',{}).get('frac'), 'peak_hbm_gb', d.get('peak_hbm_gb'), d.get('last_micro_batch_ms'))
"
done
} 2>&1 | tee gpurun_out/r06/run1.txt
